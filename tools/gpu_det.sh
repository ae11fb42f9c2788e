#!/bin/bash
# deterministic mode + trajectory + new golden tests on the GPU: bash tools/gpu_det.sh TAG
TAG=${1:-r04_det}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_deterministic.py tests/test_trajectory.py tests/test_augment_golden.py tests/test_abi.py -m gpu -x -q -s > $OUT/pytest_new.log 2>&1; rc=$?
grep -E "trajectory|default mode|passed|failed|Error|error" $OUT/pytest_new.log | tail -20
cp gpurun_out/trajectory_*.json $OUT/ 2>/dev/null
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_train_bf16.json 2> $OUT/bench_train_bf16.err || { echo "bench failed"; tail -5 $OUT/bench_train_bf16.err; exit 1; }
python -c "import json; d=json.load(open('$OUT/bench_train_bf16.json')); print(d['value'], d['ms_per_step'], d.get('deterministic_mode'))"
timeout -k 10 300 python bench.py --no-cpu-baseline --deterministic > $OUT/bench_train_bf16_deterministic.json 2> $OUT/bench_det.err || { echo "det bench failed"; tail -5 $OUT/bench_det.err; exit 1; }
python -c "import json; d=json.load(open('$OUT/bench_train_bf16_deterministic.json')); print(d['value'], d['ms_per_step'], d['config']['deterministic'], d['config']['kernel_launches_per_step'])"
