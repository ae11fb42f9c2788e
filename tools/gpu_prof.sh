#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench (graph replay); usage: bash tools/gpu_prof.sh TAG [mode]
TAG=${1:-p}; MODE=${2:-train}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$MODE -- python3 $GRAFT_REPO_ROOT/bench.py --mode $MODE --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_$MODE.log 2>&1) || { echo "rocprof failed"; tail -5 $OUT/prof_$MODE.log; exit 1; }
f=$(find $OUT/prof_$MODE -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${MODE}_kernel_stats.csv && head -12 "$f" | cut -c1-160
find $OUT/prof_$MODE -name "*kernel_trace.csv" -delete
tail -2 $OUT/prof_$MODE.log | cut -c1-300
