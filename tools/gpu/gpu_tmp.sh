#!/bin/bash
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for v in 1 0 1; do echo "--- bench BIGQ=$v"; PRIMX_GEMM_BIGQ=$v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k[:24]: round(v['ms_per_step'],3) for k,v in d['kernels'].items()})"; done
