#!/bin/bash
# Builds libsmx.so (gfx950) next to the sources.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -fno-slp-vectorize: hipcc otherwise packs the butterflies into v_pk_*_f32, which issue at a
# quarter of the v_fma_f32 rate on gfx950 (measured: forward launch 140 us -> see DESIGN.md).
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -Wno-unused-result"
OUT=${SMX_OUT:-libsmx.so}
BDIR=build${SMX_TAG:-}
FLAGS="$FLAGS ${SMX_EXTRA:-}"
mkdir -p $BDIR
PIDS=""
for f in smx_decim smx_fourstep smx_fourstep2 smx_conv1 smx_direct smx_block smx_time smx_api; do
  if [ ! -f $BDIR/$f.o ] || [ $f.hip -nt $BDIR/$f.o ] || [ smx_core.h -nt $BDIR/$f.o ] \
     || [ smx_kernels.h -nt $BDIR/$f.o ] || [ smx_tables.h -nt $BDIR/$f.o ] || [ smx_launch.h -nt $BDIR/$f.o ] || [ smx_fs_big.h -nt $BDIR/$f.o ] \
     || [ ../../include/smx.h -nt $BDIR/$f.o ]; then
    rm -f $BDIR/$f.o                       # a failed compile must not leave a stale object to link
    $HIPCC $FLAGS -c $f.hip -o $BDIR/$f.o &
    PIDS="$PIDS $!"
  fi
done
for p in $PIDS; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT $BDIR/smx_decim.o $BDIR/smx_fourstep.o $BDIR/smx_fourstep2.o $BDIR/smx_conv1.o $BDIR/smx_direct.o $BDIR/smx_block.o $BDIR/smx_time.o $BDIR/smx_api.o
echo "built $(pwd)/$OUT"
