#!/bin/bash
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# CPU-baseline thread sweep (1 block, full width)
python - > gpurun_out/cpu_threads.txt 2>&1 <<'PY'
import os, sys, time, torch
sys.path.insert(0, '.')
from oracle import dit_ref, synth
dit_ref.ATTN_DTYPE = torch.float32
cfg = dict(in_channels=68, condition_channels=768, hidden_size=1152, depth=1)
sd = synth.dit_state_dict(0, **cfg)
x = synth.tensor(0, "x", (1, 2048, 68)); y = synth.tensor(0, "y", (1, 1370, 768)); t = torch.tensor([960])
print("cpu_count", os.cpu_count())
for nt in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    with torch.no_grad():
        dit_ref.dit_forward_with_cfg(sd, x, t, y, 16, 6.0)
        t0 = time.perf_counter(); dit_ref.dit_forward_with_cfg(sd, x, t, y, 16, 6.0); dt = time.perf_counter() - t0
    print(nt, "threads:", round(dt, 3), "s per block-step", flush=True)
PY
cat gpurun_out/cpu_threads.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r1 -- python bench.py --no-cpu-baseline --steps 25 > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
echo "rocprof exit $?"
ls -R gpurun_out/prof | head -20
f=$(find gpurun_out/prof -name "*kernel_stats*" | head -1); echo $f; head -40 "$f"
