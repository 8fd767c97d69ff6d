#!/bin/bash
# conv_ls_kernel timing ablations (compile-time, results are wrong): which part of a launch is loading, which MFMA, which epilogue.
#   build: for a in 1 2 3 4 5 6 7; do SRC=conv_ws tools/build_variant.sh lsab$a -DSALT_LS_ABLATE=$a; done     then (gpurun): tools/ls_ablate.sh
# SALT_LS_ABLATE bits: 1 = no fragment reads / MFMAs, 2 = no DMA, 4 = no epilogue  (7 = launch + prologue + barriers only)
cd "$(dirname "$0")/.."
V=$PWD/open-solution-salt-identification_amd/csrc/_variants
SHAPES='"32,256,16,16,256:0" "32,128,32,32,128:0" "32,320,128,128,64:0" "32,128,64,64,64:0"'
echo "== full kernel"; eval timeout 300 python tools/conv_bench.py $SHAPES 2>&1 | grep "^conv"
for a in 4 1 2 5 6 3 7; do
  echo "== SALT_LS_ABLATE=$a"; eval SALT_LIB=$V/libsaltnet_hip.lsab$a.so timeout 300 python tools/conv_bench.py $SHAPES 2>&1 | grep "^conv"
done
