#!/bin/bash
# round-end artefacts (rounds 3 - 6; ROUND=rN names the output directory): GPU tests + smoke + benches (default ddim incl. the decode leg and the timed job, decode, c4, the other
# BASELINE shapes), kernel traces of the bench commands, PMC passes (traffic: FETCH_SIZE / WRITE_SIZE; MFMA utilisation) - each PMC
# pass on its own, with --kernel-trace only
R=${ROUND:-r6}
OUT=gpurun_out/final_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -z "$ONLY_PROFILES" ]; then
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > $OUT/gpu_tests.log 2>&1; echo "pytest exit $?" >> $OUT/gpu_tests.log; tail -2 $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 | tee $OUT/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"; cut -c1-300 $OUT/bench_default.json
timeout 600 python bench.py --config decode > $OUT/bench_decode.json 2> $OUT/bench_decode.err; echo "decode exit $?"
timeout 900 python bench.py --config c4 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 exit $?"
: > $OUT/bench_other.jsonl
timeout 600 python bench.py --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-decode-leg >> $OUT/bench_other.jsonl 2>/dev/null
timeout 600 python bench.py --batch 8 --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-decode-leg >> $OUT/bench_other.jsonl 2>/dev/null
timeout 600 python bench.py --batch 4 --n-prim 4096 --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-decode-leg >> $OUT/bench_other.jsonl 2>/dev/null
timeout 600 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg >> $OUT/bench_other.jsonl 2>/dev/null
fi   # (ONLY_PROFILES=1: the traces and counter passes alone)
# the profiled commands carry --no-side-legs: the batch-8 / bf16 / torch-reference legs launch the same kernel names at other shapes
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --steps 25 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace_decode -- python bench.py --config decode --no-cpu-baseline --no-parity > $OUT/bench_trace_decode.json 2> $OUT/bench_trace_decode.err
B="python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --steps 3 --warmup 1 --repeats 1 --no-kernel-events"
D="python bench.py --config decode --no-cpu-baseline --no-parity --steps 2 --warmup 1 --repeats 1 --no-kernel-events"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- $B > /dev/null 2> $OUT/fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- $B > /dev/null 2> $OUT/write.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch_decode -- $D > /dev/null 2> $OUT/fetch_decode.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write_decode -- $D > /dev/null 2> $OUT/write_decode.err
M="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --kernel-trace --pmc $M --output-format csv -d $OUT -o mfma -- $B > /dev/null 2> $OUT/mfma.err
timeout 300 rocprofv3 --kernel-trace --pmc $M --output-format csv -d $OUT -o mfma_decode -- $D > /dev/null 2> $OUT/mfma_decode.err
# the batch-8 regime (T = 32768 token rows per launch: BASELINE configs[2] / [3] per GPU) - trace, traffic, matrix-pipe utilisation with the clock
B8="python bench.py --batch 8 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT -o trace_b8 -- $B8 --steps 6 --warmup 2 > $OUT/bench_trace_b8.json 2> $OUT/bench_trace_b8.err
B8="$B8 --steps 2 --warmup 1 --repeats 1 --no-kernel-events"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch_b8 -- $B8 > /dev/null 2> $OUT/fetch_b8.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write_b8 -- $B8 > /dev/null 2> $OUT/write_b8.err
timeout 300 rocprofv3 --kernel-trace --pmc $M --output-format csv -d $OUT -o mfma_b8 -- $B8 > /dev/null 2> $OUT/mfma_b8.err
# configs[4] per GPU (bf16, N_prim = 4096, batch 4): the attention shape of VERDICT r5 item 6
B4="python bench.py --batch 4 --n-prim 4096 --dtype bf16 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --steps 2 --warmup 1 --repeats 1 --no-kernel-events"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace_c5 -- $B4 > /dev/null 2> $OUT/trace_c5.err
timeout 300 rocprofv3 --kernel-trace --pmc $M --output-format csv -d $OUT -o mfma_c5 -- $B4 > /dev/null 2> $OUT/mfma_c5.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch_c5 -- $B4 > /dev/null 2> $OUT/fetch_c5.err
f() { find $PWD/$OUT -name "$1" | head -1; }
for db in $(find $OUT -name "*.db"); do python tools/rocprof_summary.py $db ${db%.db}_summary.txt > /dev/null; done
FC=$(f fetch_counter_collection.csv); WC=$(f write_counter_collection.csv); FD=$(f fetch_decode_counter_collection.csv); WD=$(f write_decode_counter_collection.csv)
MC=$(f mfma_counter_collection.csv); MD=$(f mfma_decode_counter_collection.csv); O=$PWD/$OUT
F8=$(f fetch_b8_counter_collection.csv); W8=$(f write_b8_counter_collection.csv); M8=$(f mfma_b8_counter_collection.csv)
M5=$(f mfma_c5_counter_collection.csv); F5=$(f fetch_c5_counter_collection.csv)
(cd tools && python pmc_traffic.py $FC $WC $O/traffic_ddim.json > $O/traffic_ddim.txt; python pmc_traffic.py $FD $WD $O/traffic_decode.json > $O/traffic_decode.txt
 python pmc_mfma_util.py $MC - $O/mfma_util_ddim.txt > /dev/null; python pmc_mfma_util.py $MD - $O/mfma_util_decode.txt > /dev/null
 python pmc_traffic.py $F8 $W8 $O/traffic_b8.json b8 > $O/traffic_b8.txt
 python pmc_mfma_util.py $M8 - $O/mfma_util_b8.txt b8 > /dev/null
 python pmc_mfma_util.py $M5 - $O/mfma_util_c5.txt c5 > /dev/null
 python pmc_any.py $F5 --mode=c5 attn > $O/fetch_c5_attn.txt)
head -14 $OUT/mfma_util_ddim.txt | cut -c1-170; head -12 $OUT/mfma_util_b8.txt | cut -c1-170; grep attn $OUT/mfma_util_c5.txt | cut -c1-170
# keep the raw counter tables small enough to travel: kernel names dominate their size
for c in $(find $OUT -name "*_counter_collection.csv"); do python - "$c" <<'PY'
import csv, re, sys
src = sys.argv[1]
with open(src) as f, open(src.replace(".csv", "_short.csv"), "w", newline="") as g:
    r = csv.DictReader(f)
    keep = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    w = csv.DictWriter(g, keep)
    w.writeheader()
    for row in r:
        row["Kernel_Name"] = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])[:70]
        w.writerow({k: row[k] for k in keep})
PY
rm -f "$c"; done
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -size +20M -delete
du -sh $OUT; ls $OUT | head -60
