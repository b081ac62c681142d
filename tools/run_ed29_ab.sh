# A/B of the two limb forms of the Curve25519 hand-scheduled kernels: Edwards tests, scalar-mul / generator-mul timing, per-kernel durations
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_edwards.py -m gpu -x -q 2>&1 | tail -3
for L in 32 29; do ARKMPC_ED_LIMBS=$L MSM_LOG2N=10 python tools/ed_bench.py 2>&1 | grep -E "scalar-muls|generator muls" | tee gpurun_out/ed_bench_limbs$L.jsonl; done
R=$PWD; cd /tmp && export TMPDIR=/tmp
for L in 32 29; do
  ARKMPC_ED_LIMBS=$L MSM_LOG2N=10 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_ed$L -o p -- python $R/tools/ed_bench.py > $R/gpurun_out/ed_prof_$L.log 2>&1
  f=$(find /tmp/prof_ed$L -name '*kernel_stats.csv' | head -1)
  grep -E "k_ed_smul|k_ed_gen_chain" "$f" | cut -c1-200 | tee $R/gpurun_out/ed_kernel_stats_limbs$L.csv
done
