#!/bin/bash
# conv_glds_kernel vs conv_mfma_kernel on the layer shapes of the R34 hypercolumn net (forward micro-benchmark)
cd "$(dirname "$0")/.."
for shape in "32 64 64 64 64" "32 128 32 32 128" "32 256 16 16 256" "32 512 8 8 512" "32 320 128 128 64" "32 64 128 128 32" "32 32 128 128 64" "32 768 8 8 512" "32 192 32 32 128" "32 128 64 64 64"; do
  SALT_CONV_V2=0 python tools/conv_micro.py $shape 3 1 bf16 50 0 2>/dev/null | tail -1
  for cfg in 6 7 8; do
    python tools/conv_micro.py $shape 3 1 bf16 50 $cfg 2>/dev/null | tail -1
  done
done
