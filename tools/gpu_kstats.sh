#!/bin/bash
# rocprofv3 kernel stats of the default bench, filtered: bash tools/gpu_kstats.sh TAG "grep-pattern"
TAG=$1; PAT=${2:-.}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof.log 2>&1) || { echo "rocprof failed"; tail -5 $OUT/prof.log; exit 1; }
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" -delete
grep -E "$PAT" $OUT/kernel_stats.csv | cut -c1-200 | head -40
