#!/bin/bash
export LEDN_EXPERIMENTAL=1   # (this script sets A/B knobs: led-net_amd/_env.py)
# cost experiments of stem_wgrad_reg_kernel (the simplest of the wave-private-LDS kernels): conv3x3.hip rebuilt ON THE BOX with
# -DLEDN_SW_EXP=n (1 no matrix instructions, 2 no LDS tile, 3 no patch-row arithmetic; results WRONG by construction).
TAG=${1:-expsw}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py --incremental > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
cd led-net_amd/csrc
for e in ${EXPS:-1 2 3}; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLEDN_SW_EXP=$e -c conv3x3.hip -o /tmp/conv3x3_$e.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libledn_sw$e.so $(ls build/*.o | grep -v conv3x3.o) /tmp/conv3x3_$e.o ) &
done; wait; cd ../..
echo "== baseline"; timeout -k 10 300 python tools/stem_bench.py 2>&1 | grep "^stem wgrad"
for e in ${EXPS:-1 2 3}; do echo "== LEDN_SW_EXP=$e"; LEDN_HIP_LIB=/tmp/libledn_sw$e.so timeout -k 10 300 python tools/stem_bench.py 2>&1 | grep "^stem wgrad"; done
