#!/bin/bash
# round 6: attention workgroup order - query tiles fastest inside an XCD (new) against (batch, head) fastest (libprimx_head.so): timings, the batch-8 step,
# and the fabric traffic (FETCH_SIZE) of the batch-8 launches
OUT=gpurun_out/r6_attn_order
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$PWD/3dtopia-xl_amd/csrc
timeout 600 python -m pytest tests/test_hip_attention.py tests/test_hip_dit.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -2
for v in head hip head hip; do
  echo "== $v"
  PRIMX_LIB=$L/libprimx_$v.so timeout 300 python tools/attn_bench.py 2>&1 | grep TFLOP
  PRIMX_LIB=$L/libprimx_$v.so timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --steps 6 --warmup 2 --no-kernel-events 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v batch8 ms_per_step', round(d['ms_per_step'],3), [round(x,3) for x in d['repeats_ms_per_step']])
"
  PRIMX_LIB=$L/libprimx_$v.so timeout 300 python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --steps 25 --no-kernel-events 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v batch1 ms_per_step', round(d['ms_per_step'],3), [round(x,3) for x in d['repeats_ms_per_step']])
"
done
for v in head hip; do
  REPS=3 PRIMX_LIB=$L/libprimx_$v.so timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch_$v -- python tools/attn_bench.py > /dev/null 2> $OUT/fetch_$v.err
  python - $(find $OUT -name "fetch_${v}_counter_collection.csv") $v <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "attn_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        acc[int(r["Grid_Size"]) // 512].append(float(r["Counter_Value"]))
for g, v in sorted(acc.items()):
    print(sys.argv[2], "attn_kernel", g, "workgroups: FETCH_SIZE", round(sum(v) / len(v)), "KiB -> x2 (gfx950) =", round(2 * sum(v) / len(v) * 1024 / 1e6, 1), "MB per launch, n =", len(v))
PY
done
rm -rf $OUT
