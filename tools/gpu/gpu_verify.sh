#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/gpu_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/gpu_tests.log; tail -3 gpurun_out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench exit $?"; cat gpurun_out/bench_default.json
# the kernel-selection switches: the GEMM / DiT / VAE parity tests under each alternate path
if [ "$1" = "--switches" ]; then
  for kv in PRIMX_GEMM_KT32=1 PRIMX_GEMM_KT64_MIN=257 PRIMX_DIT_BLOCKS_CALL=0 PRIMX_GEMM_LOADER=0 PRIMX_GEMM_NOBIG=1 PRIMX_GEMM_BIGHEADS_MIN=0 PRIMX_GEMM_NOGEMV=1 PRIMX_WPREFETCH=0 PRIMX_WPREFETCH=1 PRIMX_NULL_KV_DEDUP=0 PRIMX_CFG_STREAMS=1 PRIMX_DIT_FUSE_LN=0 PRIMX_DIT_FOLD=0 PRIMX_DIT_LN_TAIL=1 PRIMX_LN_FUSE=0 PRIMX_GEMM_XCD2D=0 PRIMX_CONV_REG=0; do
    echo "== $kv"; env $kv timeout 600 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py tests/test_hip_vae.py -q -x -p no:cacheprovider 2>&1 | tail -1
  done
fi
