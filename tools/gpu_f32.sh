#!/bin/bash
# f32 (reference-precision) bench lines with the per-entry kernel table: bash tools/gpu_f32.sh TAG
TAG=${1:-r04_f32}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
set -o pipefail
LEDN_BENCH_VERBOSE=60 timeout -k 10 500 python bench.py --dtype f32 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_train_f32.json 2> $OUT/bench_train_f32.err || { echo "f32 train failed"; tail -5 $OUT/bench_train_f32.err; exit 1; }
cat $OUT/bench_train_f32.json; grep "ms/step" $OUT/bench_train_f32.err > $OUT/train_f32_kernel_table.txt
LEDN_BENCH_VERBOSE=40 timeout -k 10 300 python bench.py --dtype f32 --mode infer --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_infer_f32.json 2> $OUT/bench_infer_f32.err || { echo "f32 infer failed"; tail -5 $OUT/bench_infer_f32.err; exit 1; }
cat $OUT/bench_infer_f32.json; grep "ms/step" $OUT/bench_infer_f32.err > $OUT/infer_f32_kernel_table.txt
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_train_bf16.json 2> $OUT/bench_train_bf16.err || { echo "bf16 train failed"; exit 1; }
cat $OUT/bench_train_bf16.json
