# per-kernel durations of the BN254 MSM at 2^LOG2N points (default 20) under the kernel trace
mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
LOG2N=${LOG2N:-20} SKIP_NAIVE=1 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_msm -o p -- python $R/tools/msm_bench.py > $R/gpurun_out/msm_prof.log 2>&1
f=$(find /tmp/prof_msm -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY' | tee $R/gpurun_out/msm_kernel_stats.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %4s avg_us %10.1f total_ms %8.3f  %5s%%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
grep msm_ms $R/gpurun_out/msm_prof.log
