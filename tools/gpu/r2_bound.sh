#!/bin/bash
# bound probes of the 128x144 LDS-DMA kernel: PRIMX_GEMM_PROF=1 full, 2 DMA only, 3 DMA + LDS fragment reads (no MFMAs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for P in 1 2 3; do
echo "== PROF=$P"
PRIMX_GEMM_PROF=$P timeout 120 python - <<'P' 2>&1 | grep "gemm144_dma" | tail -4
import torch, sys
sys.path.insert(0, '.')
import __graft_entry__; __graft_entry__.build()
from topia_xl_amd import ops
dev='cuda:0'
for K in (1152, 4608):
    M,N=4096,1152
    A=torch.randn(M,K,device=dev).half(); W=(torch.randn(N,K,device=dev)*0.03).half(); b=torch.randn(N,device=dev).half()
    x=torch.zeros(M,N,device=dev); gate=torch.randn(2,N,device=dev).half()
    for _ in range(2): ops.linear_gate_residual(A,W,b,gate,x,2048)
    torch.cuda.synchronize()
P
done
