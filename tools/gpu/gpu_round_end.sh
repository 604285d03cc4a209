#!/bin/bash
# round-end artefacts: GPU tests + smoke + benches (default ddim, decode, c4), kernel traces of the bench commands, PMC traffic passes
OUT=gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/gpu_tests.log 2>&1; echo "pytest exit $?" >> $OUT/gpu_tests.log; tail -2 $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"; cat $OUT/bench_default.json | cut -c1-600
timeout 600 python bench.py --config decode > $OUT/bench_decode.json 2> $OUT/bench_decode.err; echo "decode exit $?"
timeout 900 python bench.py --config c4 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 exit $?"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --no-cpu-baseline --no-parity --steps 25 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace_decode -- python bench.py --config decode --no-cpu-baseline --no-parity > $OUT/bench_trace_decode.json 2> $OUT/bench_trace_decode.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python bench.py --no-cpu-baseline --no-parity --steps 3 --warmup 1 --repeats 1 --no-kernel-events > /dev/null 2> $OUT/fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python bench.py --no-cpu-baseline --no-parity --steps 3 --warmup 1 --repeats 1 --no-kernel-events > /dev/null 2> $OUT/write.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch_decode -- python bench.py --config decode --no-cpu-baseline --no-parity --steps 2 --warmup 1 --repeats 1 --no-kernel-events > /dev/null 2> $OUT/fetch_decode.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write_decode -- python bench.py --config decode --no-cpu-baseline --no-parity --steps 2 --warmup 1 --repeats 1 --no-kernel-events > /dev/null 2> $OUT/write_decode.err
for db in $OUT/*.db; do python tools/rocprof_summary.py $db ${db%.db}_summary.txt; done
ls -la $OUT | head -40
