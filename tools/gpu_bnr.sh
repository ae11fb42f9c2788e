#!/bin/bash
export LEDN_EXPERIMENTAL=1   # (this script sets A/B knobs: led-net_amd/_env.py)
# BatchNorm-backward reduce kernel under its launch knobs, per shape (rocprofv3 kernel times of tools/stream_bench.py):
#   bash tools/gpu_bnr.sh TAG "LEDN_BNR_CAP=1024" "LEDN_BNR_CAP=1024 LEDN_BNR_CONTIG=1" ...   ('-' = default)
TAG=${1:-bnr}; shift; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
i=0
for V in "$@"; do
  i=$((i+1)); [ "$V" = "-" ] && V=""
  for tok in $V; do export $tok; done
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof$i -- python3 $GRAFT_REPO_ROOT/tools/stream_bench.py --only bn_act_bwd --iters 20 > $OUT/v$i.log 2>&1) || { echo "variant $i failed"; tail -5 $OUT/v$i.log; exit 1; }
  for tok in $V; do unset ${tok%%=*}; done
  f=$(find $OUT/prof$i -name "*kernel_trace.csv" | head -1)
  echo "variant $i [$V]"
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# consecutive launches of the same kernel+grid = one case; print the median duration per (kernel, grid) in order
runs = []
for r in rows:
    n = r['Kernel_Name']
    if 'bn_reduce' not in n and 'bn_apply' not in n:
        continue
    key = (n.split('(')[0][-44:], r.get('Grid_Size', r.get('Grid_Size_X', '')))
    if not runs or runs[-1][0] != key and not (len(runs) > 1 and runs[-2][0] == key):
        runs.append((key, []))
    tgt = runs[-1] if runs[-1][0] == key else runs[-2]
    tgt[1].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for (n, g), v in runs:
    v.sort()
    print(f'   {n:46s} grid {g:>9s}  n {len(v):3d}  median {v[len(v)//2]/1e3:7.1f} us  min {v[0]/1e3:7.1f}')
PY
  grep GB/s $OUT/v$i.log
  rm -rf $OUT/prof$i
done
