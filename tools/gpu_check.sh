#!/bin/bash
# One GPU-box visit: parity tests, smoke, short benches, kernel-trace profile.
# Usage (from the repo root on the GPU box): bash tools/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 $OUT/smoke.log
for DT in f32 bf16; do
  LEDN_BENCH_VERBOSE=1 timeout 900 python bench.py --mode infer --steps 5 --warmup 2 --dtype $DT > $OUT/bench_infer_$DT.json 2> $OUT/bench_infer_$DT.err
  echo "bench infer $DT rc=$?"; cat $OUT/bench_infer_$DT.json; head -30 $OUT/bench_infer_$DT.err
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_infer_bf16 -- python3 bench.py --mode infer --steps 5 --warmup 2 --dtype bf16 --no-cpu-baseline > $OUT/prof_infer.log 2>&1
echo "rocprof rc=$?"
find $OUT/prof_infer_bf16 -name "*kernel_stats*" | head -3
f=$(find $OUT/prof_infer_bf16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
