#!/bin/bash
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export REPS=5
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc -o a1 -- python tools/attn_bench.py > gpurun_out/pmc/a1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc -o a2 -- python tools/attn_bench.py > gpurun_out/pmc/a2.log 2>&1
echo done
