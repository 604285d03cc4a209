#!/bin/bash
# usage: tools/build_attn_variant.sh <name> <NW> <NSTAGE> <SPREAD>  -> 3dtopia-xl_amd/csrc/libprimx_<name>.so  (A/B via PRIMX_LIB)
set -e
cd "$(dirname "$0")/../3dtopia-xl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -I../../include \
  -DPRIMX_ATTN_NW=$2 -DPRIMX_ATTN_NSTAGE=$3 -DPRIMX_ATTN_SPREAD=$4 -c attention.hip -o attention_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libprimx_$1.so rowops.o gemm.o attention_$1.o vae.o
echo built libprimx_$1.so
