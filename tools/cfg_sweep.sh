#!/bin/bash
# conv tile-config sweep on the network's main shapes (forward, eval epilogue): usage tools/cfg_sweep.sh
for shape in "32 64 64 64 64" "32 64 128 128 64" "32 128 32 32 128" "32 256 16 16 256" "32 512 8 8 512" "32 320 128 128 64" "32 128 64 64 64" "32 64 128 128 32" "32 32 128 128 64" "32 192 32 32 128" "32 320 16 16 256" "32 768 8 8 512"; do
  for cfg in 1 2 5 4; do python tools/conv_micro.py $shape 3 1 bf16 20 $cfg 2>&1 | tail -1; done
done
