#!/bin/bash
# One GPU-box visit for the judged artefacts: default bench (with cpu_baseline), inference benches (configs B, E),
# rocprofv3 kernel-trace stats of the bench command, graph-replay timeline, PMC traffic + MFMA-busy passes.
# usage: bash tools/gpu_round.sh TAG [pytest]      (steps are chained: a failed GPU step stops the visit)
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
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_train_bf16 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_train.log 2>&1) || { echo "rocprof failed"; tail -5 $OUT/prof_train.log; exit 1; }
f=$(find $OUT/prof_train_bf16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/train_bf16_kernel_stats.csv && head -12 "$f" | cut -c1-160
find $OUT/prof_train_bf16 -name "*kernel_trace.csv" -delete
bash tools/gpu_trace.sh $TAG train > /dev/null 2>&1; head -3 $OUT/timeline_train.txt
bash tools/gpu_trace.sh $TAG infer > /dev/null 2>&1; head -3 $OUT/timeline_infer.txt
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
