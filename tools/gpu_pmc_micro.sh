#!/bin/bash
# PMC counters of one micro-benchmark shape: bash tools/gpu_pmc_micro.sh TAG "<binary> <args>" COUNTER...
TAG=$1; CMD=$2; shift 2; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && timeout -k 5 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc -- $CMD > $OUT/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
python - "$f" <<PY
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:70]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k,v in acc.items():
    if "fillBuffer" in k or "pack_weights" in k: continue
    print(k, "x", len(n[k]), {c: round(x/len(n[k])) for c,x in v.items()})
PY
rm -rf $OUT/pmc
