#!/bin/bash
# round 5: s_setprio of the loader waves (lp1, lp3) / of the compute waves (cp1) of the loader-wave GEMM kernels - builds through PRIMX_LIB, same box
OUT=gpurun_out/r5_prio
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=3dtopia-xl_amd/csrc
B="--no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 20 --warmup 5"
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), [round(x,3) for x in d["repeats_ms_per_step"]])'
for i in 1 2; do
  for v in hip lp1 lp3 cp1; do
    PRIMX_LIB=$L/libprimx_$v.so timeout 300 python bench.py $B 2>/dev/null | python -c "$J" "step $v" | tee -a $OUT/prio.txt
  done
done
for v in hip lp1 lp3 cp1; do
  echo "== $v" | tee -a $OUT/prio.txt
  PRIMX_LIB=$L/libprimx_$v.so REPS=30 timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $OUT/prio.txt
done
