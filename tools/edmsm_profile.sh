cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_edmsm -o ed -- env MSM_LOG2N=20 python $GRAFT_REPO_ROOT/tools/ed_bench.py > /dev/null 2>&1
python - <<PY
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_edmsm/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-60s calls %5s avg_us %9.1f total_ms %8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
