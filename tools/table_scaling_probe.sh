mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
for LG in 16 17 18 19; do
  LOG2N=$LG rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_t$LG -o p -- python $R/tools/ec_bench.py > /tmp/log_$LG.txt 2>&1
  f=$(find /tmp/prof_t$LG -name '*kernel_stats.csv' | head -1)
  echo "LOG2N=$LG"; grep -E "k_g1_smul_(loop29|table29)" "$f" | awk -F'",' '{print substr($1,2,20), $2}' | cut -c1-120
done
