#!/bin/bash
# VGPRs / scratch / occupancy of every kernel of libisx (hipcc remarks; no GPU needed).  usage: tools/resource_usage.sh [extra -D flags]
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" \
  -Rpass-analysis=kernel-resource-usage -c -o /dev/null altair-raytracing_amd/csrc/isx_api.hip 2>&1 |
  grep "Function Name\|VGPRs:\|ScratchSize\|Occupancy" | grep -v AGPRs | sed 's/.*remark: *//; s/ \[-Rpass.*//; s/.*: //' | paste - - - - |
  awk '{printf "%-36s VGPR %-4s scratch %-4s waves/SIMD %s\n", $1, $2, $3, $4}'
