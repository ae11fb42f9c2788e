#!/bin/bash
# One visit to a multi-GPU node answers the N > 1 design questions:
#   bench.py --gpus {1,2,4,8}  x  { default (one launch stream, SyncBN + gradient all-reduce in the hipGraph),
#                                   LEDN_MULTI_COMM=1 (one communicator per branch stream, gradient exchange overlapped
#                                   with the stem's backward), --local-bn (per-rank statistics: the exchange alone),
#                                   --collectives torch (torch.distributed, eager launches) }
# and tabulates images/s, scaling efficiency against the 1-GPU line of the same variant, the size RCCL reports for
# every communicator, collectives and collective bytes per step.  Reference launcher: tools/dist_train.sh:9-18
# (torch.distributed.launch, one process per GPU), configs/_base_/default_runtime.py:5 (backend nccl = RCCL here).
#   usage: bash tools/scale_matrix.sh [OUTDIR] [STEPS]        (needs >= 2 GPUs; bench.py refuses N > device count)
OUT=${1:-gpurun_out/scale_matrix}; STEPS=${2:-20}; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
echo "GPUs on this node: $NGPU" | tee "$OUT/table.txt"
run() {   # name, N, env assignments..., -- bench flags...
  local name=$1 n=$2; shift 2
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local f="$OUT/${name}_n${n}.json"
  env "${envs[@]}" timeout -k 10 900 python bench.py --gpus "$n" --steps "$STEPS" --warmup 5 --no-cpu-baseline "$@" > "$f" 2> "$OUT/${name}_n${n}.err" \
    || { echo "$name N=$n FAILED (see $OUT/${name}_n${n}.err)" | tee -a "$OUT/table.txt"; return 0; }
}
for n in 1 2 4 8; do
  [ "$n" -le "$NGPU" ] || continue
  run default "$n" LEDN_MULTI_COMM=0 -- 
  run multicomm "$n" LEDN_EXPERIMENTAL=1 LEDN_MULTI_COMM=1 -- --collectives rccl
  run localbn "$n" LEDN_MULTI_COMM=0 -- --local-bn
  [ "$n" -gt 1 ] && run torchdist "$n" LEDN_MULTI_COMM=0 -- --collectives torch
done
python - "$OUT" <<'PY' | tee -a "$OUT/table.txt"
import glob, json, os, sys
out = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(out, '*_n*.json'))):
    name, n = os.path.basename(f)[:-5].rsplit('_n', 1)
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    rows[(name, int(n))] = d
print(f'{"variant":10s} {"N":>2s} {"img/s":>9s} {"ms/step":>8s} {"eff":>6s} {"submission":>16s} {"coll/step":>9s} {"launches":>8s} {"MB/step":>8s}  rccl_nranks_all')
for (name, n), d in sorted(rows.items()):
    base = rows.get((name, 1))
    eff = d['value'] / (n * base['value']) if base else float('nan')
    c = d['config']
    mb = (c.get('collective_bytes_per_step') or 0) / 1e6
    print(f'{name:10s} {n:2d} {d["value"]:9.1f} {d["ms_per_step"]:8.3f} {eff:6.3f} {c["submission"]:>16s} '
          f'{str(c.get("collectives_per_step")):>9s} {str(c.get("collective_launches_per_step")):>8s} {mb:8.2f}  {c.get("rccl_nranks_all")}')
PY
