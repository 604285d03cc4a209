#!/bin/bash
# round 6: full GPU suite + the default bench line on the tree with the K / V riders (ABI 25)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6_ride_bench.json 2> gpurun_out/r6_ride_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_ride_bench.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], d['repeats_ms_per_step'])
for k in ('with_batched_kv_projection','with_layernorm_launches','with_reuse_cond_kv','with_expanded_null_kv','batch8','bf16'):
    if k in d: print(k, round(d[k]['ms_per_step'],4))
print('job', d.get('measured_job',{}).get('job_ms'))
r=d['roofline']; print(r['kernel'], r['frac'], r['traffic'], r['kernels_sum_ms_per_step'])
for k,v in sorted(d['kernels'].items()): print('   ',k, round(v['ms_per_step'],4), round(1e3*v['ms_per_step']/v['launches_per_step'],2),'us', round(v['tflops'],1))
PY
