#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the rocprofv3 passes behind the round-3 roofline line.  Outputs under gpurun_out/prof_r03/.
#   trace_default/       --kernel-trace --stats of the default `python bench.py` (the command the driver runs)
#   trace_split/ pmc_{fetch,write}_split/   the split-layout loop alone: kernel stats, FETCH_SIZE / WRITE_SIZE per launch (separate passes)
#   pmc_clock/           GRBM_GUI_ACTIVE of the same loop with the kernel trace: busy cycles / kernel wall time = the clock under the profiler
#   calib_write/         probes/write_calib under --pmc WRITE_SIZE: known byte counts in the path's store patterns
# Summary: tools/summarize_prof_r03.py
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $REPO/bench.py"
A="--layout split --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-cold"
$B > $OUT/bench_default_run.json 2> $OUT/bench_default_run.log
$B --steps 20 --warmup 5 > $OUT/bench_driver_shape_run.json 2> $OUT/bench_driver_shape_run.log
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_default -o trace -- $B > $OUT/bench_default_under_rocprof.json 2> $OUT/trace_default.log
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_split -o trace -- $B $A > $OUT/bench_trace_split.json 2> $OUT/trace_split.log
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch_split -o fetch -- $B $A > $OUT/bench_fetch_split.json 2> $OUT/fetch_split.log
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write_split -o write -- $B $A > $OUT/bench_write_split.json 2> $OUT/write_split.log
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -f csv -d $OUT/pmc_clock -o clock -- $B $A > $OUT/bench_clock.json 2> $OUT/clock.log
rocprofv3 -L 2>/dev/null | grep -i -E "WRREQ|WRITE_SIZE|WRITE_REQ" | head -40 > $OUT/write_counters_available.txt
$REPO/probes/write_calib 21 20 > $OUT/write_calib_plain.jsonl 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/calib_write -o calib -- $REPO/probes/write_calib 21 20 > $OUT/write_calib_under_pmc.jsonl 2> $OUT/calib_write.log
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -f csv -d $OUT/calib_wrreq -o calib -- $REPO/probes/write_calib 21 20 > /dev/null 2> $OUT/calib_wrreq.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_ec -o ec -- env REPS=3 LOG2N=18 MSM_LOG2N=10 python $REPO/tools/ec_bench.py > $OUT/ec_bench_pmc.json 2> $OUT/pmc_ec.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_k3 -o k3 -- $B $A > /dev/null 2> $OUT/pmc_k3.log
cd $REPO
REPS=5 LOG2N=18 python tools/ec_bench.py > $OUT/ec_bench.json 2>/dev/null
python tools/ed_bench.py 2>/dev/null > $OUT/ed_bench.jsonl
SKIP_NAIVE=1 LOG2N=10,14,18,20,22 python tools/msm_bench.py 2>/dev/null > $OUT/msm_bench.jsonl
python tools/host_mode_bench.py 2>/dev/null > $OUT/host_mode.jsonl
bash tools/host_latency.sh > $OUT/host_latency.jsonl 2>&1
python tools/kernel_suite.py 2>/dev/null | grep -v "^\[" > $OUT/kernel_suite.txt
for d in "0,0" "0,0,0,0" "0,0,0,0,0,0,0,0"; do
  N=$(echo $d | tr ',' '\n' | wc -l)
  python bench.py --single-process --gpus $N --devices $d --log2n 20 --steps 50 --warmup 5 >> $OUT/bench_single_process.jsonl 2>> $OUT/bench_single_process.log
done
python tools/summarize_prof_r03.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +8M -delete
find $OUT -name "*.db" -delete
du -sh $OUT; tail -50 $OUT/summary.txt
