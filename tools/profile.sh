#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + HBM traffic counters for bench.py.
# usage: tools/profile.sh <tag> [bench args...]     outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 200 --warmup 20 --no-cpu-baseline $*"
# pass 1: per-kernel durations
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
# pass 2/3: HBM traffic counters, each in its own run (TCC slots: FETCH_SIZE=3, WRITE_SIZE=2)
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o fetch -- python $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o write -- python $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.log
cd $REPO
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only small files for the merge back
find $OUT -name "*.csv" -size +20M -delete
ls -la $OUT $OUT/trace 2>/dev/null | head -40
