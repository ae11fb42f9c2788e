#!/bin/bash
# Second half of the judged artefacts of a round (the first half is tools/gpu_round.sh; one gpurun call each: the two
# together exceed the 20-minute limit of a call): PMC traffic + MFMA-busy passes, the published-size variant, the conv
# micro-benchmark, parity numbers, deterministic-mode and f32 lines, inference rocprof stats.
# usage: bash tools/gpu_round_b.sh TAG
TAG=${1:-r04}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
set -o pipefail
timeout -k 10 900 python tools/pmc_traffic.py --mode train --dtype bf16 --out $OUT/pmc_traffic_train_bf16.json || { echo "pmc train failed"; exit 1; }
timeout -k 10 600 python tools/pmc_traffic.py --mode infer --dtype bf16 --out $OUT/pmc_traffic_infer_bf16.json || { echo "pmc infer failed"; exit 1; }
# round 3 additions: the published-size variant next to the default, conv micro-benchmark, frozen-decision gradient numbers
timeout -k 10 600 python bench.py --backbone "cespb_depth=(2,3)" --no-cpu-baseline > $OUT/bench_train_bf16_cespb23.json 2> $OUT/bench_cespb23.err || { echo "bench cespb(2,3) failed"; tail -5 $OUT/bench_cespb23.err; exit 1; }
cat $OUT/bench_train_bf16_cespb23.json
timeout -k 10 300 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/conv_bench.txt || { echo "conv_bench failed"; exit 1; }
timeout -k 10 600 python -m pytest tests/test_train_frozen.py tests/test_bf16_blocks.py tests/test_tools.py tests/test_trajectory.py tests/test_deterministic.py -m gpu -q -s 2>&1 | grep -E "frozen step|head parameters|worst rel-L2|resume:|replay\(new|tools/train.py [0-9]|fan-in chains|trajectory|default mode|passed|failed" > $OUT/parity_numbers.txt; cat $OUT/parity_numbers.txt
cp gpurun_out/trajectory_*.json $OUT/ 2>/dev/null
# round 4 additions: deterministic-mode line, reference-precision (f32) lines with their kernel tables, inference rocprof stats
timeout -k 10 300 python bench.py --no-cpu-baseline --deterministic > $OUT/bench_train_bf16_deterministic.json 2> $OUT/bench_det.err || { echo "bench deterministic failed"; tail -5 $OUT/bench_det.err; exit 1; }
cat $OUT/bench_train_bf16_deterministic.json | cut -c1-300
LEDN_BENCH_VERBOSE=40 timeout -k 10 500 python bench.py --dtype f32 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_train_f32.json 2> $OUT/bench_train_f32.err || { echo "f32 train failed"; tail -5 $OUT/bench_train_f32.err; exit 1; }
grep "ms/step" $OUT/bench_train_f32.err > $OUT/f32_kernel_table.txt; cut -c1-200 $OUT/bench_train_f32.json
LEDN_BENCH_VERBOSE=20 timeout -k 10 300 python bench.py --dtype f32 --mode infer --no-cpu-baseline --steps 20 --warmup 3 > $OUT/bench_infer_f32.json 2> $OUT/bench_infer_f32.err || { echo "f32 infer failed"; tail -5 $OUT/bench_infer_f32.err; exit 1; }
echo "--- inference" >> $OUT/f32_kernel_table.txt; grep "ms/step" $OUT/bench_infer_f32.err >> $OUT/f32_kernel_table.txt; cut -c1-200 $OUT/bench_infer_f32.json
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_infer_bf16 -- python3 $GRAFT_REPO_ROOT/bench.py --mode infer --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_infer.log 2>&1) || { echo "rocprof infer failed"; tail -5 $OUT/prof_infer.log; exit 1; }
f=$(find $OUT/prof_infer_bf16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/infer_bf16_kernel_stats.csv && head -6 "$f" | cut -c1-160
find $OUT/prof_infer_bf16 -name "*kernel_trace.csv" -delete
