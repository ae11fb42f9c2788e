#!/bin/bash
# A/B of an environment knob on the inference bench (config B) on one box: GPU tests of the eval path first, then 4 alternating runs.
# usage: gpurun -- bash tools/gpu_ab_infer_env.sh     (edit the variant list below)
export LEDN_EXPERIMENTAL=1
python __graft_entry__.py --incremental > /dev/null 2>&1
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "blocks or parity_argmax or segmentor or mfaf" 2>&1 | tail -2
for V in "" "LEDN_POOL_PYRAMID=0" "" "LEDN_POOL_PYRAMID=0"; do
  env $V timeout -k 10 300 python bench.py --mode infer --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$V]', d['value'], d['ms_per_step'])"
done
