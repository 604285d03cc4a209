#!/bin/bash
# round 6, call 1: (A) which kernel does the vendor library run at T = 32768 and what do its counters say (clock, MFMA busy, L2), next to
# (B) ours; (C) tracked rocprof summaries of the batch-8 step
OUT=gpurun_out/r6_s1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
M="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
G="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
L="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE"
timeout 200 python tools/blas_b8.py 2>&1 | grep -v amdgpu.ids | tee $OUT/blas.txt
ONLY=32768 timeout 200 python tools/gemm_bench_big.py 2>&1 | grep TFLOP | tee $OUT/ours.txt
REPS=4 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o blas_trace -- python tools/blas_b8.py > /dev/null 2> $OUT/blas_trace.err
for p in M G L; do
  REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc ${!p} --output-format csv -d $OUT -o blas_$p -- python tools/blas_b8.py > /dev/null 2> $OUT/blas_$p.err
  ONLY=32768 REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc ${!p} --output-format csv -d $OUT -o ours_$p -- python tools/gemm_bench_big.py > /dev/null 2> $OUT/ours_$p.err
done
# (C) the batch-8 step
B="python bench.py --batch 8 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace_b8 -- $B --steps 6 --warmup 2 > $OUT/bench_trace_b8.json 2> $OUT/bench_trace_b8.err
B="$B --steps 2 --warmup 1 --repeats 1 --no-kernel-events"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch_b8 -- $B > /dev/null 2> $OUT/fetch_b8.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write_b8 -- $B > /dev/null 2> $OUT/write_b8.err
timeout 300 rocprofv3 --kernel-trace --pmc $M --output-format csv -d $OUT -o mfma_b8 -- $B > /dev/null 2> $OUT/mfma_b8.err
for db in $(find $OUT -name "*.db"); do python tools/rocprof_summary.py $db ${db%.db}_summary.txt > /dev/null; done
for c in $(find $OUT -name "*_counter_collection.csv"); do python - "$c" <<'PY'
import csv, re, sys
src = sys.argv[1]
with open(src) as f, open(src.replace(".csv", "_short.csv"), "w", newline="") as g:
    r = csv.DictReader(f)
    keep = ["Dispatch_Id", "Grid_Size", "Workgroup_Size", "Kernel_Name", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    w = csv.DictWriter(g, keep, extrasaction="ignore")
    w.writeheader()
    for row in r:
        row["Kernel_Name"] = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])[:110]
        w.writerow({k: row.get(k, "") for k in keep})
PY
rm -f "$c"; done
find $OUT -name "*.db" -size +20M -delete
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
du -sh $OUT; find $OUT -type f | head -80
grep -h "Cijk\|gemm" $(find $OUT -name "blas_trace_kernel_stats.csv") | cut -c1-300 | head
