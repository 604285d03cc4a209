#!/bin/bash
# VAE kernels (conv_in, single-read GroupNorm, staged GEMM epilogue) + PMC traffic passes of the default bench
mkdir -p gpurun_out/r2b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_vae.py tests/test_hip_gemm.py tests/test_hip_dinov2.py tests/test_hip_e2e.py tests/test_hip_dit.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r2b/tests.log 2>&1; tail -3 gpurun_out/r2b/tests.log
timeout 600 python bench.py --config decode --steps 10 > gpurun_out/r2b/bench_decode.json 2> gpurun_out/r2b/bench_decode.err; echo "decode exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r2b/bench_decode.json"))
print(d["ms_per_step"], d["value"], d.get("parity"))
for k,v in d["kernels"].items(): print("   ", k, round(v["ms_per_step"],4), v["tflops"] and round(v["tflops"],1))
PY
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2b -o fetch -- python bench.py --no-cpu-baseline --no-parity --steps 3 --warmup 1 --repeats 1 --no-kernel-events > /dev/null 2> gpurun_out/r2b/fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2b -o write -- python bench.py --no-cpu-baseline --no-parity --steps 3 --warmup 1 --repeats 1 --no-kernel-events > /dev/null 2> gpurun_out/r2b/write.err
ls gpurun_out/r2b | head -30
F=$(ls gpurun_out/r2b/*fetch*counter_collection.csv | head -1); W=$(ls gpurun_out/r2b/*write*counter_collection.csv | head -1)
head -2 $F
python tools/pmc_traffic.py $F $W gpurun_out/r2b/traffic.json
rm -f gpurun_out/r2b/*kernel_trace.csv gpurun_out/r2b/*agent_info.csv
