#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the rocprofv3 passes behind the bench line of the current round.  ROUND=r06 tools/profile.sh
# Outputs under gpurun_out/prof_$ROUND/ (scratch); `tools/profile.sh collect` (run locally afterwards) copies the summaries into profiles/$ROUND*.
#   bench_default_run.json / bench_driver_shape_run.json     the default `python bench.py` and the driver's shape (--steps 20 --warmup 5): the compact line;
#   bench_default_detail.json / bench_driver_shape_detail.json   every leg's full record of the same runs
#   trace_default/       --kernel-trace --stats of the default bench command
#   trace_split/ pmc_{fetch,write}_split/   the split-layout loop alone: kernel stats, FETCH_SIZE / WRITE_SIZE per launch (separate passes)
#   trace_aos/   pmc_{fetch,write}_aos/     the same for the arkworks AoS layout
#   pmc_clock/           GRBM_GUI_ACTIVE of the split loop with the kernel trace: busy cycles / kernel wall time = the clock under the profiler
#   pmc_k3/ pmc_ec/      VALU issue counters of K1 / K2+K3 and of the scalar-mul kernels (config 4)
#   e2e_trace/           --kernel-trace --memory-copy-trace of `bench.py --only-e2e` (the streaming host-to-host sessions)
#   bench_circuit_run.json / circuit_trace/     `bench.py --only-circuit` (resident operands, triples from host memory) and its kernel / copy trace
#   bench_group_e2e.jsonl                       `bench.py --single-process --only-e2e` with 1, 2, 4, 8 members sharing device 0 (2^20 gates in total)
# Summary: tools/summarize_prof.py
set -u
ROUND=${ROUND:-r06}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$ROUND
if [ "${1:-}" = "collect" ]; then
  set -e
  S=$OUT
  mkdir -p profiles/$ROUND profiles/${ROUND}_host profiles/${ROUND}_ec profiles/${ROUND}_e2e
  cp $S/summary.txt $S/bench_default_run.json $S/bench_driver_shape_run.json $S/bench_default_under_rocprof.json $S/clock_effect.json profiles/$ROUND/
  cp $S/bench_default_detail.json $S/bench_driver_shape_detail.json profiles/$ROUND/
  [ -f $S/bench_single_process.jsonl ] && cp $S/bench_single_process.jsonl profiles/$ROUND/
  for f in bench_circuit_run.json bench_group_e2e.jsonl; do [ -f $S/$f ] && cp $S/$f profiles/${ROUND}_e2e/; done
  [ -f $S/circuit_trace/circuit_kernel_stats.csv ] && cp $S/circuit_trace/circuit_kernel_stats.csv $S/circuit_trace/circuit_memory_copy_stats.csv profiles/${ROUND}_e2e/ 2>/dev/null
  cp $S/trace_default/trace_kernel_stats.csv profiles/$ROUND/trace_default_kernel_stats.csv
  cp $S/trace_split/trace_kernel_stats.csv profiles/$ROUND/trace_split_kernel_stats.csv
  cp $S/trace_aos/trace_kernel_stats.csv profiles/$ROUND/trace_aos_kernel_stats.csv
  cp $S/traffic_split.json profiles/traffic_split.json
  [ -f $S/traffic_aos.json ] && cp $S/traffic_aos.json profiles/traffic_aos.json
  cp $S/host_latency.jsonl $S/host_mode.jsonl $S/kernel_suite.txt profiles/${ROUND}_host/
  cp $S/ec_bench.json $S/ed_bench.jsonl $S/msm_bench.jsonl profiles/${ROUND}_ec/
  cp $S/e2e_trace/summary.json $S/e2e_trace/e2e_memory_copy_stats.csv $S/e2e_trace/e2e_kernel_stats.csv $S/e2e_trace/bench_e2e_under_rocprof.json $S/bench_e2e_run.json \
     $S/pcie_probe.jsonl $S/h2d_ramp.jsonl profiles/${ROUND}_e2e/
  python3 - $S/e2e_trace/e2e_memory_copy_trace.csv profiles/${ROUND}_e2e/one_session_copy_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
wins, cur = [], [rows[0]]
for r in rows[1:]:
    if "HOST_TO_DEVICE" not in r["Direction"]:
        continue
    if int(r["Start_Timestamp"]) - max(int(x["End_Timestamp"]) for x in cur) > 2_000_000:
        wins.append(cur); cur = []
    cur.append(r)
wins.append(cur)
sess = [w for w in wins if len(w) == 22]                 # one ISOLATED party session: x, y, a + 6 chunks of b + c + 12 peer chunks of uploads
w = sess[-1] if sess else max(wins, key=len)
t0 = int(w[0]["Start_Timestamp"])
wr = csv.writer(open(sys.argv[2], "w"))
wr.writerow(["direction", "start_ms", "end_ms", "duration_ms"])
for r in w:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wr.writerow([r["Direction"].replace("MEMORY_COPY_", ""), "%.4f" % ((a - t0) / 1e6), "%.4f" % ((b - t0) / 1e6), "%.4f" % ((b - a) / 1e6)])
PY
  ls -la profiles/$ROUND profiles/${ROUND}_host profiles/${ROUND}_ec profiles/${ROUND}_e2e
  exit 0
fi
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $REPO/bench.py"
A="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-cold"
# (bench.py ends stdout with ONE compact line; every leg's full record goes to --detail-file)
$B --detail-file $OUT/bench_default_detail.json > $OUT/bench_default_run.json 2> $OUT/bench_default_run.log
$B --steps 20 --warmup 5 --detail-file $OUT/bench_driver_shape_detail.json > $OUT/bench_driver_shape_run.json 2> $OUT/bench_driver_shape_run.log
$B --only-e2e > $OUT/bench_e2e_run.json 2> $OUT/bench_e2e_run.log
$B --only-circuit > $OUT/bench_circuit_run.json 2> $OUT/bench_circuit_run.log
for d in "0" "0,0" "0,0,0,0" "0,0,0,0,0,0,0,0"; do
  N=$(echo $d | tr ',' '\n' | wc -l)
  $B --single-process --gpus $N --devices $d --only-e2e --e2e-log2n $((20 - (N > 1 ? (N > 2 ? (N > 4 ? 3 : 2) : 1) : 0))) >> $OUT/bench_group_e2e.jsonl 2>> $OUT/bench_group_e2e.log
done
mkdir -p $OUT/circuit_trace
rocprofv3 --kernel-trace --memory-copy-trace --stats -f csv -d $OUT/circuit_trace -o circuit -- $B --only-circuit --circuit-depth 4 > $OUT/circuit_trace/bench_circuit_under_rocprof.json 2> $OUT/circuit_trace/rocprof.log
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_default -o trace -- $B --detail-file $OUT/bench_default_under_rocprof_detail.json > $OUT/bench_default_under_rocprof.json 2> $OUT/trace_default.log
for L in split aos; do
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_$L -o trace -- $B --layout $L $A > $OUT/bench_trace_$L.json 2> $OUT/trace_$L.log
  rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch_$L -o fetch -- $B --layout $L $A > $OUT/bench_fetch_$L.json 2> $OUT/fetch_$L.log
  rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write_$L -o write -- $B --layout $L $A > $OUT/bench_write_$L.json 2> $OUT/write_$L.log
done
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -f csv -d $OUT/pmc_clock -o clock -- $B --layout split $A > $OUT/bench_clock.json 2> $OUT/clock.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_ec -o ec -- env REPS=3 LOG2N=18 MSM_LOG2N=10 python $REPO/tools/ec_bench.py > $OUT/ec_bench_pmc.json 2> $OUT/pmc_ec.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_k3 -o k3 -- $B --layout split $A > /dev/null 2> $OUT/pmc_k3.log
mkdir -p $OUT/e2e_trace
rocprofv3 --kernel-trace --memory-copy-trace --stats -f csv -d $OUT/e2e_trace -o e2e -- $B --only-e2e > $OUT/e2e_trace/bench_e2e_under_rocprof.json 2> $OUT/e2e_trace/rocprof.log
cd $REPO
python tools/summarize_e2e_trace.py $OUT/e2e_trace > $OUT/e2e_trace/summary.json 2>&1
[ -x probes/pcie_probe ] && probes/pcie_probe > $OUT/pcie_probe.jsonl 2>&1
[ -x probes/h2d_ramp_probe ] && probes/h2d_ramp_probe > $OUT/h2d_ramp.jsonl 2>&1
REPS=5 LOG2N=18 python tools/ec_bench.py > $OUT/ec_bench.json 2>/dev/null
python tools/ed_bench.py 2>/dev/null > $OUT/ed_bench.jsonl
SKIP_NAIVE=1 LOG2N=10,14,18,20,22 python tools/msm_bench.py 2>/dev/null > $OUT/msm_bench.jsonl
python tools/host_mode_bench.py 2>/dev/null > $OUT/host_mode.jsonl
bash tools/host_bench.sh latency > $OUT/host_latency.jsonl 2>&1
python tools/kernel_suite.py 2>/dev/null | grep -v "^\[" > $OUT/kernel_suite.txt
for d in "0,0" "0,0,0,0" "0,0,0,0,0,0,0,0"; do
  N=$(echo $d | tr ',' '\n' | wc -l)
  python bench.py --single-process --gpus $N --devices $d --log2n 20 --steps 50 --warmup 5 >> $OUT/bench_single_process.jsonl 2>> $OUT/bench_single_process.log
done
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
