#!/bin/bash
# one-pass BatchNorm backward: parity, micro-benchmark, whole-step A/B.  bash tools/gpu_bnf.sh TAG
export LEDN_EXPERIMENTAL=1
TAG=${1:-r04_bnf}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_stream_fast.py -m gpu -x -q -k fused > $OUT/pytest_bnf.log 2>&1; rc=$?
tail -5 $OUT/pytest_bnf.log
[ $rc = 0 ] || exit $rc
timeout -k 10 200 python tools/bn_fused_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bn_fused_bench.txt || exit 1
for i in 1 2; do
  for f in 0 1; do
    LEDN_BN_FUSED=$f timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 > $OUT/bench_f${f}_$i.json 2> $OUT/bench_f${f}_$i.err || { echo "bench fused=$f failed"; tail -5 $OUT/bench_f${f}_$i.err; exit 1; }
    python -c "import json; d=json.load(open('$OUT/bench_f${f}_$i.json')); print('LEDN_BN_FUSED=$f', d['value'], d['ms_per_step'], d['config']['kernel_launches_per_step'])"
  done
done
