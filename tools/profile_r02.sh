#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): every rocprofv3 pass behind the round-2 numbers.  Outputs under gpurun_out/prof_r02/.
#   trace_default/   --kernel-trace --stats of the default `python bench.py` (headline + AoS + config 4 + config 5 legs in one run)
#   pmc_{fetch,write}_{split,aos}/   FETCH_SIZE / WRITE_SIZE per launch, each counter in its own run (TCC slots), kernel-trace only
#   pmc_ec/          VALU / wave / stall counters of the scalar-mul kernels (config 4)
# Summaries: tools/summarize_prof_r02.py (writes profiles/-shaped text + json next to the raw csv).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r02
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $REPO/bench.py"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_default -o trace -- $B > $OUT/bench_default.json 2> $OUT/trace_default.log
for layout in split aos; do
  A="--layout $layout --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_$layout -o trace -- $B $A > $OUT/bench_trace_$layout.json 2> $OUT/trace_$layout.log
  rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch_$layout -o fetch -- $B $A > $OUT/bench_fetch_$layout.json 2> $OUT/fetch_$layout.log
  rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write_$layout -o write -- $B $A > $OUT/bench_write_$layout.json 2> $OUT/write_$layout.log
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_ec -o ec -- env REPS=3 LOG2N=18 python $REPO/tools/ec_bench.py > $OUT/ec_bench_pmc.json 2> $OUT/pmc_ec.log
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_ec -o trace -- env REPS=5 LOG2N=18 python $REPO/tools/ec_bench.py > $OUT/ec_bench_trace.json 2> $OUT/trace_ec.log
cd $REPO
REPS=5 LOG2N=18 python tools/ec_bench.py > $OUT/ec_bench.json 2>/dev/null
./probes/mulrate 2000 > $OUT/mulrate.jsonl 2>&1
python tools/genmul_bench.py > $OUT/genmul_bench.json 2>/dev/null
python tools/ed_bench.py > $OUT/ed_bench.json 2>/dev/null
SKIP_NAIVE=1 LOG2N=10,12,14,16,18,20,22 python tools/msm_bench.py 2>/dev/null > $REPO/gpurun_out/msm_bench_r02.jsonl
python tools/kernel_suite.py 2>/dev/null | grep -v "^\[" > $OUT/kernel_suite.txt
FID=1 python tools/kernel_suite.py 2>/dev/null | grep -v "^\[" > $OUT/kernel_suite_bls12_381.txt
ARKMPC_MOCK_LINK=device ./ark-mpc_amd/lib/arkmpc_host_bench point_batch_mul 262144 3 > $OUT/host_point_batch_mul.jsonl 2>&1
ARKMPC_POINT_MUL_LITERAL=1 ARKMPC_MOCK_LINK=device ./ark-mpc_amd/lib/arkmpc_host_bench point_batch_mul 262144 3 >> $OUT/host_point_batch_mul.jsonl 2>&1
bash tools/host_bench.sh > $OUT/host_bench.jsonl 2>&1
python bench.py --layout aos --no-extras > $OUT/bench_aos.json 2>/dev/null
python tools/summarize_prof_r02.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +8M -delete
find $OUT -name "*.db" -delete
du -sh $OUT; tail -60 $OUT/summary.txt
