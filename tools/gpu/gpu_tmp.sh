#!/bin/bash
mkdir -p gpurun_out/rm
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d gpurun_out/rm -o p1 -- python tools/raymarch_bench.py > gpurun_out/rm/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/rm -o p2 -- python tools/raymarch_bench.py > gpurun_out/rm/p2.log 2>&1
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/rm/**/p?_counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'raymarch_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print({k: f"{sum(v)/len(v):.4g}" for k, v in acc.items()})
PY
