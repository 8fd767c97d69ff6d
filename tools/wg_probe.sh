#!/bin/bash
# weight-gradient micro-benchmark over the layer shapes of the ResNet34 U-Net: fast kernel vs the generic one
for g in 0 1; do
for shape in "32 256 16 16 256" "32 64 64 64 64" "32 512 8 8 512" "32 128 32 32 128" "32 64 128 128 64"; do
if [ $g = 1 ]; then export SALT_WGRAD_GENERIC=1; fi
python tools/wgrad_micro.py $shape 2>&1 | grep "conv_wgrad\|wgrad_reduce" | sed "s/^/generic=$g /"
done; done
