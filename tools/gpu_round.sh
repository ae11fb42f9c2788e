#!/bin/bash
# One GPU-box visit for the judged artefacts: default bench (with cpu_baseline), infer bench,
# rocprofv3 kernel-trace stats of the bench command, PMC traffic passes.
# usage: bash tools/gpu_round.sh TAG [pytest]      (steps are chained: a failed GPU step stops the visit)
TAG=${1:-r01}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
set -o pipefail
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ "$2" = pytest ]; then
  timeout -k 10 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest_gpu.log
  [ $rc = 0 ] || exit $rc
fi
LEDN_BENCH_VERBOSE=60 timeout -k 10 600 python bench.py > $OUT/bench_train_bf16.json 2> $OUT/bench_train_bf16.err || { echo "bench train failed"; tail -5 $OUT/bench_train_bf16.err; exit 1; }
cat $OUT/bench_train_bf16.json; grep -v "amdgpu.ids\|Warning\|warn" $OUT/bench_train_bf16.err | head -50
LEDN_BENCH_VERBOSE=30 timeout -k 10 600 python bench.py --mode infer > $OUT/bench_infer_bf16.json 2> $OUT/bench_infer_bf16.err || { echo "bench infer failed"; tail -5 $OUT/bench_infer_bf16.err; exit 1; }
cat $OUT/bench_infer_bf16.json; grep -v "amdgpu.ids\|Warning\|warn" $OUT/bench_infer_bf16.err | head -24
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_train_bf16 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_train.log 2>&1) || { echo "rocprof failed"; tail -5 $OUT/prof_train.log; exit 1; }
f=$(find $OUT/prof_train_bf16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/train_bf16_kernel_stats.csv && head -30 "$f" | cut -c1-200
find $OUT/prof_train_bf16 -name "*kernel_trace.csv" -delete
timeout -k 10 900 python tools/pmc_traffic.py --mode train --dtype bf16 --out $OUT/pmc_traffic_train_bf16.json || { echo "pmc failed"; exit 1; }
