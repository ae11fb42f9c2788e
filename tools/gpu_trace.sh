#!/bin/bash
# kernel trace with timestamps + timeline analysis; usage: bash tools/gpu_trace.sh TAG [mode]
TAG=${1:-t}; MODE=${2:-train}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_$MODE -- python3 $GRAFT_REPO_ROOT/bench.py --mode $MODE --steps 6 --warmup 2 --no-cpu-baseline --trace-only > $GRAFT_REPO_ROOT/$OUT/trace_$MODE.log 2>&1) || { echo "rocprof failed"; tail -5 $OUT/trace_$MODE.log; exit 1; }
f=$(find $OUT/trace_$MODE -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py "$f" | tee $OUT/timeline_$MODE.txt
rm -rf $OUT/trace_$MODE
