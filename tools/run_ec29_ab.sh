# A/B of the two limb forms of the BN254 hand-scheduled kernels: curve + MSM tests in both forms, config-4 / MSM / generator-mul timing
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_curve.py tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -3
ARKMPC_EC_LIMBS=32 timeout 1500 python -m pytest tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -2
for L in 32 29; do
  ARKMPC_EC_LIMBS=$L python tools/ec_bench.py 2>&1 | tail -1 | tee gpurun_out/ec_bench_$L.json
  ARKMPC_EC_LIMBS=$L LOG2N=16,18,20,22 SKIP_NAIVE=1 python tools/msm_bench.py 2>&1 | grep msm_ms | tee gpurun_out/msm_bench_$L.jsonl
  ARKMPC_EC_LIMBS=$L python tools/genmul_bench.py 2>&1 | grep generator | tee gpurun_out/genmul_bench_$L.jsonl
done
