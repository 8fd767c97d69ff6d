#!/bin/bash
cd "$(dirname "$0")/.."
V=$PWD/open-solution-salt-identification_amd/csrc/_variants
SHAPES='"32,256,16,16,256:0" "32,128,32,32,128:0" "32,320,128,128,64:0" "32,128,64,64,64:0" "32,512,8,8,512:0"'
for lib in default lsn6 lsn8; do
  echo "== $lib"; if [ $lib = default ]; then unset SALT_LIB; else export SALT_LIB=$V/libsaltnet_hip.$lib.so; fi
  eval timeout 300 python tools/conv_bench.py $SHAPES 2>&1 | grep "^conv"
  timeout 600 python -m pytest tests/test_gpu_conv_ws.py -m gpu -q -x -k "ls or planar" 2>&1 | tail -1
done
unset SALT_LIB
bash tools/ab_libs.sh default open-solution-salt-identification_amd/csrc/_variants/libsaltnet_hip.lsn8.so open-solution-salt-identification_amd/csrc/_variants/libsaltnet_hip.lsn6.so
