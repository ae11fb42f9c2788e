#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench (train, bf16): per-kernel table into gpurun_out/$TAG/
# usage: bash tools/gpu_stats.sh TAG [extra bench args]
TAG=${1:-st}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1) || { echo "rocprof failed"; tail -5 $OUT/prof.log; exit 1; }
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
t=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_timeline.py "$t" > $OUT/timeline.txt 2>&1
find $OUT/prof -name "*kernel_trace.csv" -delete
grep '"metric"' $OUT/prof.log | tail -1 > $OUT/bench.json
head -60 $OUT/timeline.txt
