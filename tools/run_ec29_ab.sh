set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_curve.py -m gpu -x -q 2>&1 | tail -15
ARKMPC_EC_LIMBS=32 python tools/ec_bench.py 2>&1 | tail -1 | tee gpurun_out/ec_bench_32.json
python tools/ec_bench.py 2>&1 | tail -1 | tee gpurun_out/ec_bench_29.json
