#!/bin/bash
# A/B of library builds on the block calls inside ONE gpurun call: tools/ab_block.sh libA.so libB.so ...
C=$PWD/tensor-cuda-fft-_amd/csrc
for round in 1 2 3; do
  for lib in "$@"; do
    SMX_LIB=$C/$lib timeout -k 10 120 python tools/block_kbench.py --shape ${SHAPE:-64x4096x256x128} 2>/dev/null | grep blk_fwd
  done
done
