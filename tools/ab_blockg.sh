#!/bin/bash
# A/B of library builds on the whole block step (hipGraph, fwd+bwd): tools/ab_blockg.sh libA.so libB.so ...
C=$PWD/tensor-cuda-fft-_amd/csrc
for round in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib "; SMX_LIB=$C/$lib timeout -k 10 120 python tools/block_bench.py --only fused_block --shape ${SHAPE:-64x4096x256x128} 2>/dev/null | grep fused
  done
done
