#!/bin/bash
# round 4, session 7: per-kernel same-box comparison of the round-3 tree (prev_tree/) and the round-4 tree (HIP events per launch,
# bench.py's `kernels` record), after the K-tail predicate of the generic GEMM kernel became a uniform branch; suites that touch it
OUT=gpurun_out/r4_s7
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_gemm.py tests/test_hip_vae.py tests/test_hip_dit.py tests/test_hip_dinov2.py -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1; echo "suites: $(tail -1 $OUT/tests.log)"
H="bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg"
for rep in 1 2; do
  (cd prev_tree && timeout 300 python $H > ../$OUT/k_r3_$rep.json 2>> ../$OUT/err.txt)
  timeout 300 python $H --no-side-legs > $OUT/k_r4_$rep.json 2>> $OUT/err.txt
done
python - <<'PY'
import json
out = "gpurun_out/r4_s7/"
def load(f):
    r = json.load(open(out + f)); return r["ms_per_step"], r["kernels"]
for rep in (1, 2):
    a, ka = load(f"k_r3_{rep}.json"); b, kb = load(f"k_r4_{rep}.json")
    print(f"rep {rep}: step r3 {a:.3f} ms | r4 {b:.3f} ms")
    norm = lambda k: k.replace("2, 2, 2, 2, 0, 0>", "2, 2, 2, 2, 0>")
    ka = {norm(k): v for k, v in ka.items()}
    for k in sorted(set(ka) | set(kb)):
        x, y = ka.get(k), kb.get(k)
        print(f"   {k[:70]:70s} r3 {x['ms_per_step'] if x else float('nan'):7.3f} ({x['launches_per_step'] if x else 0:5.1f})  r4 {y['ms_per_step'] if y else float('nan'):7.3f} ({y['launches_per_step'] if y else 0:5.1f})")
PY
