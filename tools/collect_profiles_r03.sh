#!/bin/bash
# Copies the summaries of tools/profile_r03.sh (gpurun_out/prof_r03, scratch) into profiles/ (tracked): what DESIGN.md / bench.py cite.
set -e
S=gpurun_out/prof_r03
mkdir -p profiles/r03 profiles/r03_host profiles/r03_ec
cp $S/summary.txt $S/bench_default_run.json $S/bench_driver_shape_run.json $S/bench_default_under_rocprof.json $S/clock_effect.json $S/write_calib.json \
   $S/write_counters_available.txt $S/bench_single_process.jsonl profiles/r03/
cp $S/trace_default/trace_kernel_stats.csv profiles/r03/trace_default_kernel_stats.csv
cp $S/trace_split/trace_kernel_stats.csv profiles/r03/trace_split_kernel_stats.csv
cp $S/traffic_split.json profiles/traffic_split.json
cp $S/host_latency.jsonl $S/host_mode.jsonl $S/kernel_suite.txt profiles/r03_host/
cp $S/ec_bench.json $S/ed_bench.jsonl $S/msm_bench.jsonl profiles/r03_ec/
ls -la profiles/r03 profiles/r03_host profiles/r03_ec
