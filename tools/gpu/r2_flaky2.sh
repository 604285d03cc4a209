#!/bin/bash
# repeat runs of the suites that go through the loader-wave GEMM and the timestep plan (intermittent ordering bugs would show here)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  timeout 200 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py tests/test_hip_fullconfig.py -m gpu -q --tb=short -x -p no:cacheprovider 2>&1 | tail -1
done
python - <<'P'
import torch, sys
sys.path.insert(0, '.')
import __graft_entry__; __graft_entry__.build()
from topia_xl_amd import ops
dev='cuda:0'
torch.manual_seed(0)
M,N,K=4096,1152,4608
A=torch.randn(M,K,device=dev).half(); W=(torch.randn(N,K,device=dev)*0.03).half(); b=torch.randn(N,device=dev).half()
gate=torch.randn(2,N,device=dev).half()
ref=None
bad=0
for it in range(300):
    x=torch.ones(M,N,device=dev)
    ops.linear_gate_residual(A,W,b,gate,x,2048)
    if ref is None: ref=x.clone()
    elif not torch.equal(x,ref): bad+=1
print("300 repeats of the K=4608 gate-residual GEMM, mismatching runs:", bad)
P
