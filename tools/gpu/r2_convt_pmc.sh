#!/bin/bash
mkdir -p gpurun_out/pmc_convt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d gpurun_out/pmc_convt -o f -- python bench.py --config decode --no-cpu-baseline --no-parity --steps 2 --warmup 1 --repeats 1 --no-kernel-events > /dev/null 2> gpurun_out/pmc_convt/err.log
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_convt/**/f_counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        if 'convt' in k or 'conv3' in k or 'groupnorm' in k:
            print(k, {c: round(sum(v)/len(v)/1024, 1) for c, v in d.items()}, 'MiB (FETCH x2 for bytes)')
PY
