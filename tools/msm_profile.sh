mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
LOG2N=20 SKIP_NAIVE=1 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_msm -o p -- python $R/tools/msm_bench.py > $R/gpurun_out/msm_prof.log 2>&1
f=$(find /tmp/prof_msm -name '*kernel_stats.csv' | head -1)
head -16 "$f" | cut -c1-150 | tee $R/gpurun_out/msm_kernel_stats.csv
tail -2 $R/gpurun_out/msm_prof.log | cut -c1-300
