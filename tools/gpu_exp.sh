#!/bin/bash
export LEDN_EXPERIMENTAL=1   # (this script sets A/B knobs: led-net_amd/_env.py)
# phase-cost experiments of conv_mfma_kernel: the library rebuilt ON THE BOX with -DLEDN_EXP=n (1 no stores, 2 no matrix
# phase, 3 no fetch, 4 no fetch / commit; results are WRONG by construction: timing only).  bash tools/gpu_exp.sh TAG
TAG=${1:-exp}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
cd led-net_amd/csrc
for e in ${EXPS:-1 2 3 4}; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLEDN_EXP=$e -c conv_mfma.hip -o /tmp/conv_mfma_$e.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libledn_exp$e.so $(ls build/*.o | grep -v conv_mfma.o) /tmp/conv_mfma_$e.o ) &
done; wait; cd ../..
echo "== baseline"; timeout -k 10 300 python tools/conv_bench.py --k ${KS:-3} 2>&1 | grep -v amdgpu.ids | tee $OUT/base.txt || exit 1
for e in ${EXPS:-1 2 3 4}; do
  echo "== LEDN_EXP=$e"; LEDN_HIP_LIB=/tmp/libledn_exp$e.so timeout -k 10 300 python tools/conv_bench.py --k ${KS:-3} 2>&1 | grep -v amdgpu.ids | tee $OUT/exp$e.txt || exit 1
done
