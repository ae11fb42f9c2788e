#!/bin/bash
# the round-end check as the driver runs it: all GPU tests, smoke, default bench
TAG=${1:-full}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
grep "\[build\]" $OUT/build.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $OUT/pytest_gpu.log | cut -c1-300
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python __graft_entry__.py --incremental smoke > $OUT/smoke.log 2>&1 || { tail -5 $OUT/smoke.log; exit 1; }; tail -2 $OUT/smoke.log
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }; cat $OUT/bench.json
