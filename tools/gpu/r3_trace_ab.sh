#!/bin/bash
# per-kernel durations inside the step for two libraries (rocprofv3 kernel trace of the same bench command)
OUT=gpurun_out/tr
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CS=$PWD/3dtopia-xl_amd/csrc
for v in ${LIBS:-prev hip}; do
  PRIMX_LIB=$CS/libprimx_$v.so timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o tr_$v -- python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events --steps 10 --warmup 3 --repeats 2 > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  db=$(find $OUT -name "tr_${v}_results.db" | head -1); python tools/rocprof_summary.py $db $OUT/tr_${v}_summary.txt > /dev/null; echo "== $v"; sed -n 3,9p $OUT/tr_${v}_summary.txt | cut -c1-110; grep -A8 "per (kernel, grid)" $OUT/tr_${v}_summary.txt | cut -c1-110
done
find $OUT -name "*.db" -size +30M -delete
