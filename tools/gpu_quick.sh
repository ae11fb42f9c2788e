#!/bin/bash
# short GPU visit: selected tests + bf16 train/infer benches (no CPU baseline).
# usage: bash tools/gpu_quick.sh tag "pytest -k expr" [table-lines]
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "${2:-mfma or bf16}" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $OUT/pytest.log
[ $rc = 0 ] || exit $rc
for MODE in train infer; do
  LEDN_BENCH_VERBOSE=400 timeout -k 10 600 python bench.py --mode $MODE --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$MODE.json 2> $OUT/bench_$MODE.err || { echo "bench $MODE failed"; tail -5 $OUT/bench_$MODE.err; exit 1; }
  echo "bench $MODE"; cat $OUT/bench_$MODE.json; grep "ms/step" $OUT/bench_$MODE.err | head -${3:-40}
done
