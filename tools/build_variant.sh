#!/bin/bash
# usage: tools/build_variant.sh <source stem (gemm, attention, conv3, ...)> <name> <extra -D / -mllvm flags...>
#   -> 3dtopia-xl_amd/csrc/libprimx_<name>.so with that one object rebuilt with the extra flags (same-box A/B via PRIMX_LIB)
set -e
cd "$(dirname "$0")/../3dtopia-xl_amd/csrc"
stem=$1; name=$2; shift; shift
extra=""
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 -I../../include $extra "$@" -c $stem.hip -o ${stem}_$name.o
objs=""
for o in rowops gemm attention vae primsdf raymarch fp32 conv3 conv3s8 conv3s8c32 convt dit_host; do
  if [ $o = $stem ]; then objs="$objs ${stem}_$name.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libprimx_$name.so $objs
echo built libprimx_$name.so
