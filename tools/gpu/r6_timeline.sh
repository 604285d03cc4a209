cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kt in 0 1; do
PRIMX_GEMM_KT32=$kt PRIMX_GEMM_PROF=1 REPS=2 ONLY=32768 timeout 200 python tools/gemm_bench_big.py 2>&1 | grep "gemm288q_dma<" | awk 'NR%5==0' | cut -c1-330
done
