#!/bin/bash
# One GPU-box visit: parity tests, smoke, short benches, kernel-trace profile.
# Usage (from the repo root on the GPU box): bash tools/gpu_check.sh [tag] [quick]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
timeout 600 python __graft_entry__.py --incremental smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -6 $OUT/smoke.log
for MODE in train infer; do
  for DT in bf16 f32; do
    LEDN_BENCH_VERBOSE=1 timeout 1200 python bench.py --mode $MODE --steps 4 --warmup 2 --dtype $DT > $OUT/bench_${MODE}_$DT.json 2> $OUT/bench_${MODE}_$DT.err
    echo "bench $MODE $DT rc=$?"; cat $OUT/bench_${MODE}_$DT.json; grep -v amdgpu.ids $OUT/bench_${MODE}_$DT.err | head -40
  done
done
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_bf16 -- python3 bench.py --mode train --steps 3 --warmup 1 --dtype bf16 --no-cpu-baseline > $OUT/prof_train.log 2>&1
echo "rocprof rc=$?"
f=$(find $OUT/prof_train_bf16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-220
