# PMC pass over the 29-bit scalar-mul kernels (config 4): instruction counts, wave cycles, wait cycles -- separate run, counters only
mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d /tmp/pmc1 -o ec -- env REPS=1 LOG2N=18 python $R/tools/ec_bench.py > /tmp/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY -f csv -d /tmp/pmc2 -o ec -- env REPS=1 LOG2N=18 python $R/tools/ec_bench.py > /tmp/pmc2.log 2>&1
python3 - <<'PY' | tee $R/gpurun_out/ec29_pmc.txt
import csv, glob, collections
for d in ("/tmp/pmc1", "/tmp/pmc2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "smul" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k in acc:
            print(k, {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
