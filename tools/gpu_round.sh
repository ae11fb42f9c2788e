#!/bin/bash
# One GPU-box visit for the judged artefacts: default bench (with cpu_baseline), inference benches (configs B, E),
# rocprofv3 kernel-trace stats of the bench command, graph-replay timeline, PMC traffic + MFMA-busy passes.
# usage: bash tools/gpu_round.sh TAG [pytest]      (steps are chained: a failed GPU step stops the visit)
# second half of the visit (PMC passes, variants, parity numbers, deterministic / f32 lines): tools/gpu_round_b.sh TAG
TAG=${1:-r04}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
set -o pipefail
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ "$2" = pytest ]; then
  timeout -k 10 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest_gpu.log
  [ $rc = 0 ] || exit $rc
fi
LEDN_BENCH_VERBOSE=80 timeout -k 10 600 python bench.py > $OUT/bench_train_bf16.json 2> $OUT/bench_train_bf16.err || { echo "bench train failed"; tail -5 $OUT/bench_train_bf16.err; exit 1; }
cat $OUT/bench_train_bf16.json; grep "ms/step" $OUT/bench_train_bf16.err > $OUT/train_kernel_table.txt
LEDN_BENCH_VERBOSE=30 timeout -k 10 600 python bench.py --mode infer > $OUT/bench_infer_bf16.json 2> $OUT/bench_infer_bf16.err || { echo "bench infer failed"; tail -5 $OUT/bench_infer_bf16.err; exit 1; }
cat $OUT/bench_infer_bf16.json
timeout -k 10 600 python bench.py --mode infer --batch 4 --height 1024 --width 2048 --no-cpu-baseline > $OUT/bench_infer_bf16_configE.json 2> $OUT/bench_infer_configE.err || { echo "bench config E failed"; tail -5 $OUT/bench_infer_configE.err; exit 1; }
cat $OUT/bench_infer_bf16_configE.json
timeout -k 10 600 python bench.py --collectives rccl --no-cpu-baseline > $OUT/bench_train_bf16_rccl_selftest.json 2> $OUT/bench_rccl.err || { echo "bench rccl failed"; tail -5 $OUT/bench_rccl.err; exit 1; }
cat $OUT/bench_train_bf16_rccl_selftest.json
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_train_bf16 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-det-probe > $GRAFT_REPO_ROOT/$OUT/prof_train.log 2>&1) || { echo "rocprof failed"; tail -5 $OUT/prof_train.log; exit 1; }
f=$(find $OUT/prof_train_bf16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/train_bf16_kernel_stats.csv && head -12 "$f" | cut -c1-160
find $OUT/prof_train_bf16 -name "*kernel_trace.csv" -delete
bash tools/gpu_trace.sh $TAG train > /dev/null 2>&1; head -3 $OUT/timeline_train.txt
bash tools/gpu_trace.sh $TAG infer > /dev/null 2>&1; head -3 $OUT/timeline_infer.txt
