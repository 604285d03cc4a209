#!/bin/bash
mkdir -p gpurun_out/pmc_conv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/conv_bench.py
export REPS=4
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_conv -o c1 -- python tools/conv_bench.py > gpurun_out/pmc_conv/c1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_conv -o c2 -- python tools/conv_bench.py > gpurun_out/pmc_conv/c2.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_conv -o c3 -- python tools/conv_bench.py > gpurun_out/pmc_conv/c3.log 2>&1
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_conv/**/c?_counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv3_s4c256' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-1], {k: f"{sum(v)/len(v):.4g}" for k, v in acc.items()})
PY
