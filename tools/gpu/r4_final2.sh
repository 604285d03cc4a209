#!/bin/bash
# round-4 end artefacts on the FINAL tree (after the K-tail branch of the generic GEMM kernel; the PMC tables of r4_final.sh stay valid:
# no other kernel changed): whole GPU suite, smoke, benches, kernel traces, then the switch matrix
R=r4
OUT=gpurun_out/final2_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > $OUT/gpu_tests.log 2>&1; echo "pytest exit $?" >> $OUT/gpu_tests.log; tail -2 $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 | tee $OUT/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"; cut -c1-300 $OUT/bench_default.json
timeout 600 python bench.py --config decode > $OUT/bench_decode.json 2> $OUT/bench_decode.err; echo "decode exit $?"
timeout 900 python bench.py --config c4 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 exit $?"
: > $OUT/bench_other.jsonl
X="--no-cpu-baseline --no-parity --no-decode-leg --no-side-legs"
timeout 400 python bench.py --batch 8 --steps 6 --warmup 2 $X >> $OUT/bench_other.jsonl 2>/dev/null
timeout 400 python bench.py --batch 8 --dtype bf16 --steps 6 --warmup 2 $X >> $OUT/bench_other.jsonl 2>/dev/null
timeout 400 python bench.py --batch 4 --n-prim 4096 --dtype bf16 --steps 6 --warmup 2 $X >> $OUT/bench_other.jsonl 2>/dev/null
timeout 400 python bench.py --dtype bf16 --steps 20 --warmup 5 $X >> $OUT/bench_other.jsonl 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py $X --steps 25 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace_decode -- python bench.py --config decode --no-cpu-baseline --no-parity > $OUT/bench_trace_decode.json 2> $OUT/bench_trace_decode.err
for db in $(find $OUT -name "*.db"); do python tools/rocprof_summary.py $db ${db%.db}_summary.txt > /dev/null; done
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
echo "== switch matrix"
bash tools/gpu/r4_switches.sh
du -sh $OUT
