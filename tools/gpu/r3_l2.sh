#!/bin/bash
# (1) CU ingest probe: LDS-DMA vs global->VGPR operand delivery (tools/probe/cu_ingest.hip)
# (2) L2 / fabric counters of the GEMMs inside the DiT step against the same kernels launched back to back in isolation:
#     what is different about the epilogue's stores in the step (32k cycles per workgroup against 12k isolated)?
OUT=gpurun_out/l2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 60 ./tools/probe/cu_ingest 1152; timeout 60 ./tools/probe/cu_ingest 4608) 2>&1 | tee $OUT/ingest.txt
(timeout 120 rocprofv3 -L 2>&1 || timeout 120 rocprofv3 --list-avail 2>&1) > $OUT/avail.txt
grep -o "TCC_[A-Z0-9_a-z]*\|TCP_[A-Z0-9_a-z]*" $OUT/avail.txt | sort -u > $OUT/avail_tc.txt
wc -l $OUT/avail_tc.txt
G[1]="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum"
G[2]="TCC_HIT_sum TCC_MISS_sum TCC_WRITE_sum TCC_READ_sum"
G[3]="TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_NORMAL_EVICT_sum"
G[4]="TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_32B_sum"
G[5]="TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
for i in 1 2 3 4 5; do
  names=""
  for c in ${G[$i]}; do if grep -qx "$c" $OUT/avail_tc.txt; then names="$names $c"; else echo "counter $c not available"; fi; done
  [ -z "$names" ] && continue
  echo "== group $i:$names"
  timeout 200 rocprofv3 --kernel-trace --pmc $names --output-format csv -d $OUT -o step_g$i -- python bench.py --no-cpu-baseline --no-parity --no-decode-leg --steps 3 --warmup 1 --repeats 1 --no-kernel-events > /dev/null 2> $OUT/step_g$i.err
  ONLY=fc1,proj REPS=6 timeout 200 rocprofv3 --kernel-trace --pmc $names --output-format csv -d $OUT -o iso_g$i -- python tools/gemm_bench.py > /dev/null 2> $OUT/iso_g$i.err
  for w in step iso; do
    f=$(find $OUT -name "${w}_g${i}_counter_collection.csv" | head -1)
    [ -n "$f" ] && { echo "-- $w"; python tools/pmc_any.py $f gemm ln_modulate attn | cut -c1-330; rm -f $f; }
  done
done 2>&1 | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
