#!/bin/bash
# generic GPU visit: bash tools/gpu_visit.sh TAG "pytest -k expr (or '-' for none)" [convbench] [bench]
TAG=${1:-v}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ "$2" != "-" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s -k "$2" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
  grep -E "worst|frozen step|fan-in|passed|failed|Error" $OUT/pytest.log | cut -c1-400 | tail -20
  [ $rc = 0 ] || exit $rc
fi
if [ -n "$3" ]; then timeout -k 10 300 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_bench.txt || exit 1; fi
if [ -n "$4" ]; then
  LEDN_BENCH_VERBOSE=400 timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_train.json 2> $OUT/bench_train.err || { tail -5 $OUT/bench_train.err; exit 1; }
  cat $OUT/bench_train.json; grep "ms/step" $OUT/bench_train.err | head -60
fi
if [ -n "$5" ]; then timeout -k 10 300 python tools/conv_bench.py $5 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_bench_ab.txt || exit 1; fi
