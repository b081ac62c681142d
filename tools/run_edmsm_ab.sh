# A/B of the Curve25519 MSM bucket fold: hand-scheduled 29-bit stream (default) vs the compiled kernel (ARKMPC_EDMSM_ASM=0): tests + timing
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_edwards.py -m gpu -x -q -k "msm" 2>&1 | tail -3
for M in 0 1; do ARKMPC_EDMSM_ASM=$M MSM_LOG2N=14,16,18,20 python tools/ed_bench.py 2>&1 | grep "MSM" | tee gpurun_out/edmsm_bench_asm$M.jsonl; done
