#!/bin/bash
# per-kernel VGPR / occupancy / LDS / code size of the gfx950 build
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value --cuda-device-only -S -o /tmp/hc_dev.s optimal_conv_amd/csrc/hconv.hip -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|Occupancy|LDS Size|ScratchSize" | sed -E 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - - - \
 | sed -E 's/Function Name: _Z[0-9]+//; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/; s/LDS Size \[bytes\/block\]/lds/' | awk '{printf "%-28s %s %s %s %s %s %s %s %s\n", substr($1,1,28),$2,$3,$4,$5,$6,$7,$8,$9}'
