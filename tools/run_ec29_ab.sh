# A/B of the two limb forms of the BN254 hand-scheduled kernels: curve + MSM tests, config-4 timing, per-kernel durations under the kernel trace
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_curve.py tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -3
for L in 32 29; do ARKMPC_EC_LIMBS=$L python tools/ec_bench.py 2>&1 | tail -1 | tee gpurun_out/ec_bench_$L.json; done
R=$PWD; cd /tmp && export TMPDIR=/tmp
for L in 32 29; do
  ARKMPC_EC_LIMBS=$L rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$L -o p -- python $R/tools/ec_bench.py > $R/gpurun_out/ec_prof_$L.log 2>&1
  f=$(find /tmp/prof_$L -name '*kernel_stats.csv' | head -1)
  head -4 "$f" | cut -c1-160 | tee $R/gpurun_out/ec_kernel_stats_$L.csv
done
