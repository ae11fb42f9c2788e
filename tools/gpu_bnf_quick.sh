export LEDN_EXPERIMENTAL=1
OUT=gpurun_out/r04_bnf2; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_stream_fast.py -m gpu -x -q -k fused 2>&1 | tail -2
timeout -k 10 200 python tools/bn_fused_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bn_fused_bench.txt
