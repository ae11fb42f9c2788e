#!/bin/bash
# short GPU visit: selected tests + bf16 benches.  usage: bash tools/gpu_quick.sh tag "pytest -k expr"
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -x -q -k "${2:-mfma or bf16}" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log
for MODE in train infer; do
  LEDN_BENCH_VERBOSE=400 timeout 900 python bench.py --mode $MODE --steps 4 --warmup 2 --dtype bf16 --no-cpu-baseline > $OUT/bench_$MODE.json 2> $OUT/bench_$MODE.err
  echo "bench $MODE rc=$?"; cat $OUT/bench_$MODE.json; grep -v amdgpu.ids $OUT/bench_$MODE.err | head -${3:-30}
done
timeout 600 python bench.py --mode train --steps 4 --warmup 2 --dtype bf16 --no-cpu-baseline --no-graph > $OUT/bench_train_eager.json 2> /dev/null; echo "eager:"; cat $OUT/bench_train_eager.json
