#!/bin/bash
OUT=gpurun_out/wpf3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 1 2; do
PRIMX_WPREFETCH=$m timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o tr$m -- python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events --steps 6 --warmup 2 --repeats 1 > $OUT/bench.json 2> $OUT/bench.err
db=$(find $OUT -name "tr${m}_results.db" | head -1); echo "== mode $m"; python - <<PY
import sqlite3
db = sqlite3.connect("$db")
rows = db.execute("select name, grid_x, grid_y, duration from kernels where name like '%attn_kernel%' order by start").fetchall()
cross = [r[3] for i, r in enumerate(rows) if i % 2 == 0]; selfa = [r[3] for i, r in enumerate(rows) if i % 2 == 1]
print("cross-attention avg %.2f us, self-attention avg %.2f us (%d launches each)" % (sum(cross)/len(cross)/1e3, sum(selfa)/len(selfa)/1e3, len(cross)))
ln = db.execute("select avg(duration) from kernels where name like '%ln_modulate%'").fetchone()[0]
print("LN avg %.2f us" % (ln/1e3))
PY
done
find $OUT -name "*.db" -delete
