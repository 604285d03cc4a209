#!/bin/bash
# end of round 6: the configs[1] step, the batch-8 step and the whole 25-step job of the FINAL tree against the round-5 tree on ONE box, alternating
# (prev_tree/: `git worktree add prev_tree 568bf5e` + its own build, git-ignored, travels with the snapshot) - the pool's boxes differ by up to 9 %,
# this comparison does not
OUT=gpurun_out/r6_ab_vs_r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), [round(x,4) for x in d.get('repeats_ms_per_step',[])], 'job_ms', (d.get('measured_job') or {}).get('job_ms'))"; }
B="--no-cpu-baseline --no-parity --no-side-legs --no-kernel-events --steps 20 --warmup 5"
for i in 1 2 3; do
  (cd prev_tree && timeout 300 python bench.py $B 2>/dev/null | p "r5 tree  configs[1]") | tee -a $OUT/ab.txt
  timeout 300 python bench.py $B 2>/dev/null | p "r6 final configs[1]" | tee -a $OUT/ab.txt
done
B8="--batch 8 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 6 --warmup 2"
for i in 1 2; do
  (cd prev_tree && timeout 300 python bench.py $B8 2>/dev/null | p "r5 tree  batch 8") | tee -a $OUT/ab.txt
  timeout 300 python bench.py $B8 2>/dev/null | p "r6 final batch 8" | tee -a $OUT/ab.txt
done
B5="--batch 4 --n-prim 4096 --dtype bf16 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 6 --warmup 2"
(cd prev_tree && timeout 300 python bench.py $B5 2>/dev/null | p "r5 tree  configs[4] shape") | tee -a $OUT/ab.txt
timeout 300 python bench.py $B5 2>/dev/null | p "r6 final configs[4] shape" | tee -a $OUT/ab.txt
