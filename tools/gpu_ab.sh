#!/bin/bash
export LEDN_EXPERIMENTAL=1   # (this script sets A/B knobs: led-net_amd/_env.py)
# A/B on ONE box (box-to-box spread is larger than most single changes): bench.py train under several env settings.
# usage: bash tools/gpu_ab.sh TAG "ENV1=.. ENV2=.." "ENV1=.." ...   (each argument = one variant's environment; '-' = default)
TAG=${1:-ab}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
i=0
for V in "$@"; do
  i=$((i+1)); [ "$V" = "-" ] && V=""
  ARGS=""; ENVS=""
  for tok in $V; do case $tok in --*) ARGS="$ARGS $tok";; *) ENVS="$ENVS $tok";; esac; done
  STEPS="--steps 40 --warmup 5"; case "$ARGS" in *infer*) STEPS="--steps 300 --warmup 20";; esac; case "$ARGS" in *--steps*) STEPS="";; esac
  env $ENVS timeout -k 10 400 python bench.py $STEPS --no-cpu-baseline $ARGS > $OUT/v$i.json 2> $OUT/v$i.err || { echo "variant $i ($V) failed"; tail -8 $OUT/v$i.err; exit 1; }
  echo "variant $i [$V]: $(python -c "import json,sys; d=json.load(open('$OUT/v$i.json')); print(d['value'], d['unit'], d['ms_per_step'],'ms', d['config'].get('collectives'))")"
done
