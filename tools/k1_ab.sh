#!/bin/bash
# conv micro-benchmark of library variants: tools/k1_ab.sh <variant>...   (default build first)
cd "$(dirname "$0")/.."
V=open-solution-salt-identification_amd/csrc/_variants
export SALT_CONV_V2=0
SH="32,64,64,64,64:0 32,128,32,32,128:0 32,256,16,16,256:0 32,512,8,8,512:0 32,320,128,128,64:0 32,64,128,128,32:0"
echo "== default"; python tools/conv_bench.py $SH 2>/dev/null
for d in "$@"; do
  echo "== $d"; SALT_LIB=$V/libsaltnet_hip.$d.so python tools/conv_bench.py $SH 2>/dev/null
done
