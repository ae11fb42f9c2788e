#!/bin/bash
# conv micro-benchmark (+ optional MFMA parity tests): bash tools/gpu_convbench.sh tag [test]
TAG=${1:-cb}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ -n "$2" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "mfma or bf16" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log; fi
timeout 600 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_bench.txt
