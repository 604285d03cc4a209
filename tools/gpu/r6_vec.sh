#!/bin/bash
# round 6: gemm144l gate-residual epilogues with the column vectors / row pairs fetched once per workgroup (LDS) - against the previous commit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$PWD/3dtopia-xl_amd/csrc
timeout 1200 python -m pytest tests/test_hip_gemm.py tests/test_hip_fold.py tests/test_hip_dit.py tests/test_hip_fullconfig.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
for v in ${VARIANTS:-head hip head hip}; do
  echo "== $v"
  PRIMX_LIB=$L/libprimx_$v.so timeout 300 python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --steps 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v ms_per_step', round(d['ms_per_step'],3), [round(x,3) for x in d['repeats_ms_per_step']])
for k,v in d['kernels'].items():
    if ('gemm' in k or 'attn' in k) and v['ms_per_step']>0.3: print('   ',k, round(v['ms_per_step'],4), round(1e3*v['ms_per_step']/v['launches_per_step'],2),'us', round(v['tflops'],1))
"
done
bash tools/gpu/r6_timeline144.sh | grep 144l
