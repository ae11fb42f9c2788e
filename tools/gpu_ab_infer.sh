#!/bin/bash
export LEDN_EXPERIMENTAL=1   # (this script sets A/B knobs: led-net_amd/_env.py)
# inference A/B on one box by environment: bash tools/gpu_ab_infer.sh TAG "pytest -k or -" "ENV" "ENV" ...
TAG=$1; K=$2; shift 2; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ "$K" != "-" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$K" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log | cut -c1-300
  [ $rc = 0 ] || exit $rc
fi
i=0
for V in "$@"; do
  i=$((i+1)); [ "$V" = "-" ] && V=""
  env $V timeout -k 10 400 python bench.py --mode infer --steps 300 --warmup 20 --no-cpu-baseline > $OUT/i$i.json 2> $OUT/i$i.err || { echo "variant $i failed"; tail -5 $OUT/i$i.err; exit 1; }
  echo "infer variant $i [$V]: $(python -c "import json; d=json.load(open('$OUT/i$i.json')); print(d['value'], 'img/s', d['ms_per_step'], 'ms')")"
done
