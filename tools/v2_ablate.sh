#!/bin/bash
cd "$(dirname "$0")/.."
V=open-solution-salt-identification_amd/csrc/_variants
S1="32,256,16,16,256"; S2="32,64,64,64,64"; S3="32,512,8,8,512"; S4="32,128,32,32,128"
echo "== v2 full"; python tools/conv_bench.py $S1:7 $S2:6 $S3:8 $S4:6 2>/dev/null
for d in "$@"; do
  echo "== v2 dbg$d"; SALT_LIB=$V/libsaltnet_hip.dbg$d.so python tools/conv_bench.py $S1:7 $S2:6 $S3:8 $S4:6 2>/dev/null
done
