#!/bin/bash
# Copies the summaries of gpurun_out/prof_r02 (tools/profile_r02.sh on the GPU box) into the tracked profiles/r02* directories.
set -eu
S=gpurun_out/prof_r02
mkdir -p profiles/r02 profiles/r02_aos profiles/r02_config5 profiles/r02_ec profiles/r02_host profiles/r02_msm
cp $S/summary.txt profiles/r02/summary.txt
cp $S/trace_default/trace_kernel_stats.csv profiles/r02/trace_default_kernel_stats.csv
cp $S/trace_split/trace_kernel_stats.csv profiles/r02/trace_split_kernel_stats.csv
cp $S/bench_default.json profiles/r02/bench_default_under_rocprof.json
[ -f $S/bench_default_unprofiled.json ] && cp $S/bench_default_unprofiled.json profiles/r02/bench_default_run.json
[ -f $S/bench_driver_shape.json ] && cp $S/bench_driver_shape.json profiles/r02/bench_driver_shape_run.json
cp $S/traffic_split.json profiles/traffic_split.json
cp $S/traffic_aos.json profiles/traffic_aos.json
cp $S/traffic_aos.json profiles/r02_aos/traffic_aos.json
cp $S/trace_aos/trace_kernel_stats.csv profiles/r02_aos/trace_kernel_stats.csv
cp $S/bench_aos.json profiles/r02_aos/bench_aos_run.json
grep -E '^"?Name|k_mac_check|k_mac_verify|k_share_extract|k_to_bytes_be' $S/trace_default/trace_kernel_stats.csv > profiles/r02_config5/kernel_stats_rows.csv || true
cp $S/ec_bench.json profiles/r02_ec/ec_bench.json
[ -f $S/ed_bench.json ] && cp $S/ed_bench.json profiles/r02_ec/ed_bench.json
cp $S/mulrate.jsonl profiles/r02_ec/mulrate.jsonl
cp $S/trace_ec/trace_kernel_stats.csv profiles/r02_ec/trace_kernel_stats.csv
sed -n '/scalar-mul kernels, PMC/,$p' $S/summary.txt > profiles/r02_ec/pmc_valu.txt
for f in host_bench.jsonl host_point_batch_mul.jsonl kernel_suite.txt kernel_suite_bls12_381.txt; do cp $S/$f profiles/r02_host/$f; done
[ -f $S/genmul_bench.json ] && cp $S/genmul_bench.json profiles/r02_ec/genmul_bench.json
[ -f gpurun_out/msm_bench_r02.jsonl ] && cp gpurun_out/msm_bench_r02.jsonl profiles/r02_msm/msm_bench.jsonl
[ -f gpurun_out/soak_ec.txt ] && cp gpurun_out/soak_ec.txt profiles/r02_ec/soak_ec.txt
[ -f gpurun_out/soak_asm_r02.txt ] && cp gpurun_out/soak_asm_r02.txt profiles/r02/soak_asm.txt
git status --short profiles | head -40
