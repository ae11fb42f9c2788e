#!/bin/bash
# f32 bench lines only (+ conv tests): bash tools/gpu_f32c.sh TAG
TAG=${1:-r04_f32c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_conv_f32.py tests/test_deterministic.py -m gpu -x -q > $OUT/pytest_f32.log 2>&1; rc=$?
tail -3 $OUT/pytest_f32.log
[ $rc = 0 ] || exit $rc
LEDN_BENCH_VERBOSE=40 timeout -k 10 500 python bench.py --dtype f32 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_train_f32.json 2> $OUT/bench_train_f32.err || { echo "f32 train failed"; tail -5 $OUT/bench_train_f32.err; exit 1; }
python -c "import json; d=json.load(open('$OUT/bench_train_f32.json')); print('f32 train', d['value'], d['ms_per_step'])"; grep "ms/step" $OUT/bench_train_f32.err > $OUT/train_f32_kernel_table.txt; head -50 $OUT/train_f32_kernel_table.txt | cut -c1-150
LEDN_BENCH_VERBOSE=30 timeout -k 10 300 python bench.py --dtype f32 --mode infer --no-cpu-baseline --steps 20 --warmup 3 > $OUT/bench_infer_f32.json 2> $OUT/bench_infer_f32.err || { echo "f32 infer failed"; tail -5 $OUT/bench_infer_f32.err; exit 1; }
python -c "import json; d=json.load(open('$OUT/bench_infer_f32.json')); print('f32 infer', d['value'], d['ms_per_step'])"; grep "ms/step" $OUT/bench_infer_f32.err > $OUT/infer_f32_kernel_table.txt; head -6 $OUT/infer_f32_kernel_table.txt | cut -c1-150
