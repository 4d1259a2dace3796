#!/bin/bash
# Builds libsmx.so (gfx950) next to the sources.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result"
mkdir -p build
for f in smx_decim smx_direct smx_api; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ smx_core.h -nt build/$f.o ] \
     || [ smx_kernels.h -nt build/$f.o ] || [ smx_tables.h -nt build/$f.o ] \
     || [ ../../include/smx.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
  fi
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libsmx.so build/smx_decim.o build/smx_direct.o build/smx_api.o
echo "built $(pwd)/libsmx.so"
