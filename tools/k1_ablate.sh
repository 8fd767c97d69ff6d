#!/bin/bash
# compile-time ablations of conv_mfma_kernel (SALT_K1_DBG: 1 return after the prologue, 2 no chunk loop, 4 no epilogue, 6 = launch + prologue + first loads)
cd "$(dirname "$0")/.."
V=open-solution-salt-identification_amd/csrc/_variants
export SALT_CONV_V2=0
SH="32,64,64,64,64:0 32,128,32,32,128:0 32,256,16,16,256:0 32,512,8,8,512:0 32,320,128,128,64:0"
echo "== full"; python tools/conv_bench.py $SH 2>/dev/null
for d in "$@"; do
  echo "== k1 dbg$d"; SALT_LIB=$V/libsaltnet_hip.k$d.so python tools/conv_bench.py $SH 2>/dev/null
done
