#!/bin/bash
# A/B of the pixels-per-wave of conv1x1_wgrad_reg_kernel: bash tools/gpu_w11.sh TAG
TAG=${1:-w11}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
for v in 4 8 16 4 8 16; do
  LEDN_W11_ITERS=$v timeout -k 10 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/b_$v.json 2> $OUT/b_$v.err || { tail -5 $OUT/b_$v.err; exit 1; }
  python -c "import json; d=json.loads(open('$OUT/b_$v.json').read().strip().splitlines()[-1]); print('iters', $v, d['value'], d['ms_per_step'])"
done
