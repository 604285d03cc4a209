#!/bin/bash
mkdir -p gpurun_out/pmc_xcd
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 0; do
PRIMX_GEMM_XCD2D=$v ONLY=fc1,qkv REPS=5 timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_xcd -o x$v -- python tools/gemm_bench.py > /dev/null 2> gpurun_out/pmc_xcd/err$v.log
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_xcd/**/x?_counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'gemm288q' in r['Kernel_Name']: acc[r['Grid_Size']].append(float(r['Counter_Value']))
    print(f.split('/')[-1], {g: round(2*sum(v)/len(v)/1024,1) for g, v in acc.items()}, 'MiB fetched per launch (x2 corrected), by grid')
PY
