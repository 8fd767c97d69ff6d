#!/bin/bash
cd "$(dirname "$0")/.."
V=$PWD/open-solution-salt-identification_amd/csrc/_variants
for lib in default lsd3; do
  echo "== $lib"; if [ $lib = default ]; then unset SALT_LIB; else export SALT_LIB=$V/libsaltnet_hip.$lib.so; fi
  timeout 300 python tools/conv_bench.py "32,256,16,16,256:0" "32,128,32,32,128:0" 2>&1 | grep "^conv"
  timeout 600 python -m pytest tests/test_gpu_conv_ws.py -m gpu -q -x -k "ls" 2>&1 | tail -1
done
unset SALT_LIB
bash tools/ab_libs.sh default open-solution-salt-identification_amd/csrc/_variants/libsaltnet_hip.lsd3.so
bash tools/ab_libs.sh default open-solution-salt-identification_amd/csrc/_variants/libsaltnet_hip.lsd3.so | tail -2
