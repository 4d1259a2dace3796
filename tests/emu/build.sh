#!/bin/bash
# CPU emulator of the fused kernel (test infrastructure): same phase functions, g++, no HIP.
set -euo pipefail
cd "$(dirname "$0")"
if [ ! -f libsmx_emu.so ] || [ emu_fused.cpp -nt libsmx_emu.so ] \
   || [ ../../tensor-cuda-fft-_amd/csrc/smx_core.h -nt libsmx_emu.so ] \
   || [ ../../tensor-cuda-fft-_amd/csrc/smx_tables.h -nt libsmx_emu.so ]; then
  g++ -O2 -std=c++17 -shared -fPIC -I ../../tensor-cuda-fft-_amd/csrc -o libsmx_emu.so emu_fused.cpp
fi
echo "built $(pwd)/libsmx_emu.so"
