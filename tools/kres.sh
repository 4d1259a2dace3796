#!/bin/bash
# kernel resource usage of one translation unit: tools/kres.sh smx_conv1 [filter]
cd "$(dirname "$0")/../tensor-cuda-fft-_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -Wno-unused-result ${SMX_EXTRA:-} \
  -Rpass-analysis=kernel-resource-usage -c $1.hip -o /tmp/$1.o 2>&1 \
  | grep -E "error|Function Name|VGPRs:|VGPRs Spill|ScratchSize|LDS Size" | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' \
  | paste - - - - - | grep -E "${2:-.}"
