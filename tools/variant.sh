#!/bin/bash
# tools/variant.sh NAME "-DFLAG ..." : libsmx_NAME.so = smx_decim.hip compiled with the flags + the other objects of build/
# (A/B candidates for tools/ab_inproc.py without rebuilding the six other translation units)
set -euo pipefail
cd "$(dirname "$0")/../tensor-cuda-fft-_amd/csrc"
name=$1; flags=${2:-}
mkdir -p build_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -Wno-unused-result $flags -c smx_decim.hip -o build_$name/smx_decim.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsmx_$name.so build_$name/smx_decim.o build/smx_fourstep.o build/smx_fourstep2.o build/smx_conv1.o build/smx_direct.o build/smx_block.o build/smx_time.o build/smx_api.o
echo "built libsmx_$name.so"
