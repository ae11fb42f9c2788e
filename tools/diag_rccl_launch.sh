#!/bin/bash
OUT=gpurun_out/$1; mkdir -p $OUT
run() { name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? json_bytes=$(stat -c %s $OUT/$name.json) err_lines=$(wc -l < $OUT/$name.err) last_err=$(tail -1 $OUT/$name.err | cut -c1-100)"; }
run s20_eq python bench.py --steps 20 --warmup 5 --no-cpu-baseline --collectives=rccl
run s20_sp python bench.py --steps 20 --warmup 5 --no-cpu-baseline --collectives rccl
run s10 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --collectives rccl
run s20_fh python -X faulthandler bench.py --steps 20 --warmup 5 --no-cpu-baseline --collectives rccl
