#!/bin/bash
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
echo "--- bench new"; timeout 600 python bench.py --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k: round(v['ms_per_step'],3) for k,v in d['kernels'].items()})"
echo "--- bench prev"; (cd prev_tree && timeout 600 python bench.py --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
echo "--- torchrun nproc=1"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['ms_per_step'], d['value'])"
