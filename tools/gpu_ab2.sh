#!/bin/bash
export LEDN_EXPERIMENTAL=1   # (this script sets A/B knobs: led-net_amd/_env.py)
# A/B on one box by environment: bash tools/gpu_ab2.sh TAG "pytest -k expr or -" "ENV..." "ENV..." ...
TAG=$1; K=$2; shift 2; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ "$K" != "-" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$K" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log | cut -c1-300
  [ $rc = 0 ] || exit $rc
fi
i=0
for V in "$@"; do
  i=$((i+1)); [ "$V" = "-" ] && V=""
  env $V LEDN_BENCH_VERBOSE=60 timeout -k 10 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/v$i.json 2> $OUT/v$i.err || { echo "variant $i failed"; tail -5 $OUT/v$i.err; exit 1; }
  echo "variant $i [$V]: $(python -c "import json; d=json.load(open('$OUT/v$i.json')); print(d['value'], 'img/s', d['ms_per_step'], 'ms', d['config']['kernel_launches_per_step'], 'launches')")"
done
