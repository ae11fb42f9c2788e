#!/bin/bash
export LEDN_EXPERIMENTAL=1   # (this script sets A/B knobs: led-net_amd/_env.py)
# where the SyncBN (one-rank RCCL) step differs from the plain one: bash tools/gpu_sync_gap.sh TAG
TAG=${1:-gap}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
run() { # name, env..., -- args
  name=$1; shift
  env "$@" timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $ARGS > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; exit 1; }
  python -c "import json; d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['config'].get('collective_launches_per_step'))"
}
ARGS=""; run plain A=1 || exit 1
ARGS=""; run plain_noforks LEDN_CTX_FORKS=0 || exit 1
ARGS="--collectives rccl"; run rccl A=1 || exit 1
ARGS="--collectives rccl"; run rccl_multi LEDN_MULTI_COMM=1 || exit 1
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --collectives rccl --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof.log 2>&1) || { echo "rocprof failed"; tail -5 $OUT/prof.log; exit 1; }
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats_rccl.csv
find $OUT/prof -name "*kernel_trace.csv" -delete
head -30 $OUT/kernel_stats_rccl.csv | cut -c1-160
grep -iE "nccl|rccl|AllReduce|copy" $OUT/kernel_stats_rccl.csv | cut -c1-200
