#!/bin/bash
# round-4 end artefacts with the LayerNorm fold as the default: whole GPU suite + smoke, the default bench line (incl. the decode /
# job / batch8 / bf16 / with_layernorm_launches legs), the kernel trace of the headline command, PMC passes (FETCH_SIZE / WRITE_SIZE,
# MFMA utilisation) - each PMC pass on its own, with --kernel-trace only.  (decode / c4 / other shapes: unchanged by the fold,
# profiles/r4_bench_decode.json ... stay as produced by r4_final2.sh)
OUT=gpurun_out/final3_r4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > $OUT/gpu_tests.log 2>&1; echo "pytest exit $?" >> $OUT/gpu_tests.log; tail -2 $OUT/gpu_tests.log
grep -n "FAILED" $OUT/gpu_tests.log | head
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 | tee $OUT/smoke.txt
T0=$(date +%s)
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $? wall $(( $(date +%s) - T0 )) s"; cut -c1-250 $OUT/bench_default.json
X="--no-cpu-baseline --no-parity --no-decode-leg --no-side-legs"
PRIMX_DIT_FOLD=0 timeout 200 python bench.py $X --no-kernel-events --batch 8 --steps 6 --warmup 2 > $OUT/bench_b8_unfolded.json 2> $OUT/bench_b8_unfolded.err
python -c "import json;a=json.load(open('$OUT/bench_default.json'));b=json.load(open('$OUT/bench_b8_unfolded.json'));print('headline', round(a['ms_per_step'],3), 'with LayerNorm launches', round(a['with_layernorm_launches']['ms_per_step'],3), '| batch 8 folded', round(a['batch8']['ms_per_step'],2), 'unfolded', round(b['ms_per_step'],2), '| job ms', round(a['measured_job']['job_ms'],1))"
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py $X --steps 25 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
B="python bench.py $X --steps 3 --warmup 1 --repeats 1 --no-kernel-events"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- $B > /dev/null 2> $OUT/fetch.err
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- $B > /dev/null 2> $OUT/write.err
M="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
timeout 200 rocprofv3 --kernel-trace --pmc $M --output-format csv -d $OUT -o mfma -- $B > /dev/null 2> $OUT/mfma.err
f() { find $PWD/$OUT -name "$1" | head -1; }
for db in $(find $OUT -name "*.db"); do python tools/rocprof_summary.py $db ${db%.db}_summary.txt > /dev/null; done
FC=$(f fetch_counter_collection.csv); WC=$(f write_counter_collection.csv); MC=$(f mfma_counter_collection.csv); O=$PWD/$OUT
(cd tools && python pmc_traffic.py $FC $WC $O/traffic_ddim.json > $O/traffic_ddim.txt; python pmc_mfma_util.py $MC - $O/mfma_util_ddim.txt > /dev/null)
grep "gemm\|attn" $OUT/traffic_ddim.txt | head -16
head -12 $OUT/mfma_util_ddim.txt | cut -c1-170
for c in $(find $OUT -name "*_counter_collection.csv"); do rm -f "$c"; done
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
