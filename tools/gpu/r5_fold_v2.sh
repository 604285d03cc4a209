#!/bin/bash
# round 5, call 1: the range-safe LayerNorm fold (ABI 23) on the device - the whole GPU suite, then a same-box A/B of the configs[1] step
# and the batch-8 step against the round-4 tree (prev_tree/: `git worktree add prev_tree 6b25845` + build, git-ignored, travels with the snapshot)
OUT=gpurun_out/r5_fold_v2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s -x > $OUT/gpu_tests.log 2>&1; echo "pytest exit $?" | tee -a $OUT/gpu_tests.log; tail -3 $OUT/gpu_tests.log
grep -E "^range |folded|row mean" $OUT/gpu_tests.log | head -80 > $OUT/fold_lines.txt
B="--no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 20 --warmup 5"
for i in 1 2; do
  (cd prev_tree && timeout 300 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', d['ms_per_step'], d.get('repeats_ms_per_step'))") | tee -a $OUT/ab.txt
  timeout 300 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['ms_per_step'], d.get('repeats_ms_per_step'))" | tee -a $OUT/ab.txt
done
B8="--batch 8 --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events --steps 6 --warmup 2"
(cd prev_tree && timeout 300 python bench.py $B8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old b8', d['ms_per_step'], d.get('repeats_ms_per_step'))") | tee -a $OUT/ab.txt
timeout 300 python bench.py $B8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new b8', d['ms_per_step'], d.get('repeats_ms_per_step'))" | tee -a $OUT/ab.txt
