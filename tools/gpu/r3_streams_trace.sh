#!/bin/bash
OUT=gpurun_out/st2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PRIMX_CFG_STREAMS=1 PRIMX_GEMM_BIG_MIN=112 PRIMX_GEMM_BIGHEADS_MIN=96 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o st -- python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events --steps 6 --warmup 2 --repeats 1 > $OUT/bench.json 2> $OUT/bench.err
db=$(find $OUT -name "st_results.db" | head -1)
python tools/rocprof_summary.py $db $OUT/summary.txt > /dev/null; sed -n 1,10p $OUT/summary.txt | cut -c1-120; grep "gaps between" $OUT/summary.txt
python - <<PY
import sqlite3
db = sqlite3.connect("$db")
ks = db.execute('select name, start, "end", queue_id, stream_id from kernels order by start').fetchall() if False else None
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print(cols)
q = 'queue_id' if 'queue_id' in cols else None
rows = db.execute(f'select name, start, "end"' + (f', {q}' if q else '') + ' from kernels order by start').fetchall()
# a window in the middle of the trace: 40 consecutive kernels with start / end relative, and the queue
mid = len(rows) * 3 // 4
t0 = rows[mid][1]
for r in rows[mid:mid + 36]:
    print(f"{(r[1]-t0)/1e3:9.1f} -> {(r[2]-t0)/1e3:9.1f} us  ({(r[2]-r[1])/1e3:6.1f})  q={r[3] if q else '?'}  {r[0][:60]}")
PY
find $OUT -name "*.db" -delete
