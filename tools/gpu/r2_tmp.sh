#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 3; do
PRIMX_CONVT_PROBE=$v python - <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
import __graft_entry__; __graft_entry__.build()
from topia_xl_amd import ops
P=2048
x=torch.randn(P,64,256,device="cuda",dtype=torch.float16)
wt=(torch.randn(2048,256,device="cuda",dtype=torch.float16)*0.05)
b=torch.zeros(256,device="cuda",dtype=torch.float16)
wp=ops.pack_convt_s4(wt)
for _ in range(3): ops.convtranspose_k2s2(x,wt,b,4,Wp=wp)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.convtranspose_k2s2(x,wt,b,4,Wp=wp)
e1.record(); torch.cuda.synchronize()
print("probe", os.environ["PRIMX_CONVT_PROBE"], "us", round(e0.elapsed_time(e1)/10*1000,1))
PY
done
