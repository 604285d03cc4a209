#!/bin/bash
# SQ counters of the attention kernel (tools/attn_bench.py shapes); counters only with --kernel-trace, each pass under its own timeout
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export REPS=5
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc -o a1 -- python tools/attn_bench.py > gpurun_out/pmc/a1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc -o a2 -- python tools/attn_bench.py > gpurun_out/pmc/a2.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --output-format csv -d gpurun_out/pmc -o a3 -- python tools/attn_bench.py > gpurun_out/pmc/a3.log 2>&1
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/**/a?_counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'attn_kernel' in r['Kernel_Name']:
            acc[r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
    for g, d in acc.items():
        print(f, 'grid', g, {k: f"{sum(v)/len(v):.4g}" for k, v in d.items()})
PY
