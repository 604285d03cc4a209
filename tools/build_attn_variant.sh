#!/bin/bash
# usage: tools/build_attn_variant.sh <name> <extra -D / -mllvm flags...>  -> 3dtopia-xl_amd/csrc/libprimx_<name>.so  (A/B via PRIMX_LIB)
set -e
cd "$(dirname "$0")/../3dtopia-xl_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -I../../include "$@" -c attention.hip -o attention_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libprimx_$name.so rowops.o gemm.o attention_$name.o vae.o primsdf.o raymarch.o fp32.o conv3.o conv3s8.o conv3s8c32.o convt.o
echo built libprimx_$name.so
