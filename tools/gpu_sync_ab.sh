#!/bin/bash
export LEDN_EXPERIMENTAL=1   # (this script sets A/B knobs: led-net_amd/_env.py)
# A/B of the fused pooled-context sequence under SyncBN (one-rank RCCL self-test): bash tools/gpu_sync_ab.sh TAG
TAG=${1:-sync}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "mfaf or rccl or distributed" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc = 0 ] || exit $rc
for v in 1 0 1 0; do
  LEDN_FUSE_MFAF_SYNC=$v timeout -k 10 600 python bench.py --collectives rccl --steps 20 --warmup 5 --no-cpu-baseline > $OUT/rccl_$v.json 2> $OUT/rccl_$v.err || { tail -5 $OUT/rccl_$v.err; exit 1; }
  echo "FUSE_MFAF_SYNC=$v"; python -c "import json,sys; d=json.loads(open('$OUT/rccl_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('collective_launches_per_step'))"
done
