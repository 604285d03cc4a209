#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python - <<'P'
import torch, sys
sys.path.insert(0, '.')
import __graft_entry__; __graft_entry__.build()
from topia_xl_amd import ops
dev='cuda:0'
N,K=292608,1152
W=(torch.randn(N,K,device=dev)*0.03).half(); b=torch.randn(N,device=dev).half()
for M in (1,2,4,5,8):
    A=torch.randn(M,K,device=dev).half(); out=torch.empty(M,N,device=dev,dtype=torch.float16)
    for _ in range(3): ops.linear(A,W,b,out=out)
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.linear(A,W,b,out=out)
    e.record(); torch.cuda.synchronize()
    us=s.elapsed_time(e)*1e3/20
    print(f"M={M}: {us:.1f} us  {N*K*2/us/1e6:.2f} TB/s")
P
timeout 300 python -m pytest tests/test_hip_gemm.py -m gpu -q --tb=short -x -p no:cacheprovider -k "plain or few_row" 2>&1 | tail -2
for i in 1 2; do
  PRIMX_PLAN_TIMESTEPS=0 timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('per-step', d['ms_per_step'])"
  timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('planned ', d['ms_per_step'])"
done
