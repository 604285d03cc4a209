#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_attention.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -2
for i in 1 2; do
echo "== LDS epilogue"; timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | head -8
echo "== register epilogue"; PRIMX_LIB=$GRAFT_REPO_ROOT/3dtopia-xl_amd/csrc/libprimx_noldsepi.so timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | head -8
done
