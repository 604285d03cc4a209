#!/bin/bash
# same-box A/B: current tree vs prev_tree/ (git archive of the previous commit, built in place)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -5
echo "--- new"; timeout 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu | tail -4
if [ -d prev_tree ]; then echo "--- prev"; (cd prev_tree && timeout 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu | tail -4); fi
echo "--- new"; timeout 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu | tail -4
echo "--- bench new"; timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
if [ -d prev_tree ]; then echo "--- bench prev"; (cd prev_tree && timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"); fi
