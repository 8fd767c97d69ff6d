#!/bin/bash
# A/B variant of the library whose elementwise.hip is compiled with extra flags: tools/build_variant_ew.sh <name> <flags...>
#   -> open-solution-salt-identification_amd/csrc/_variants/libsaltnet_hip.<name>.so   (run with SALT_LIB=<that path>)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
C=open-solution-salt-identification_amd/csrc
mkdir -p $C/_variants/obj_$name
objs=""
for f in runtime conv_mfma conv_ws conv_small elementwise se loss input; do
  if [ $f = elementwise ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$C -Wno-unused-value "$@" -c $C/$f.hip -o $C/_variants/obj_$name/$f.o
    objs="$objs $C/_variants/obj_$name/$f.o"
  else
    objs="$objs $C/_obj/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/_variants/libsaltnet_hip.$name.so $objs
echo built $C/_variants/libsaltnet_hip.$name.so
