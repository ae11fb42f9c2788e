#!/bin/bash
# per-kernel register / scratch / LDS usage of one HIP source: bash tools/kres.sh led-net_amd/csrc/conv_mfma.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' |
  paste - - - - - - | sed 's/Function Name: //; s/\[bytes\/lane\]//; s/\[waves\/SIMD\]//; s/\[bytes\/block\]//' | c++filt | awk '{print substr($0,1,200)}'
