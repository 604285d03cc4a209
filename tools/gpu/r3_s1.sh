#!/bin/bash
# round 3, session 1: validate the tree (conv3 accumulators in AGPRs, new parity cases), baseline numbers, the untested XLDS
# variant, clock / MFMA-utilisation counters, vendor-library reference points.
OUT=gpurun_out/s1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CS=3dtopia-xl_amd/csrc
DESEL=""; [ -f tests/golden/xl_c2_ddim25.npz ] || DESEL="--deselect tests/test_hip_fullconfig.py::test_configs1_ddim25_trajectory"
[ -f tests/golden/xl_c5.npz ] || DESEL="$DESEL --deselect tests/test_hip_fullconfig.py::test_batched_full_width_block[xl_c5-dtype3-0.05]"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s $DESEL > $OUT/gpu_tests.log 2>&1; echo "pytest exit $?" | tee -a $OUT/gpu_tests.log
grep -E "rel-L2|max-abs|passed|failed|error" $OUT/gpu_tests.log | tail -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"; cut -c1-400 $OUT/bench_default.json
# decode: round-2 library vs this tree (same box)
for lib in r2 hip; do
  PRIMX_LIB=$PWD/$CS/libprimx_$lib.so timeout 300 python bench.py --config decode --no-cpu-baseline > $OUT/bench_decode_$lib.json 2> $OUT/bench_decode_$lib.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_decode_$lib.json"))
print("decode $lib: %.3f ms" % d["ms_per_step"], {k.split("<")[0] + k.split(">")[-1]: round(v["ms_per_step"], 3) for k, v in d["kernels"].items() if v["ms_per_step"] > 0.1})
PY
done
# XLDS: correctness first, then the K sweep against the default build
PRIMX_LIB=$PWD/$CS/libprimx_xlds.so timeout 300 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py -m gpu -q -x --tb=short -p no:cacheprovider > $OUT/xlds_tests.log 2>&1; echo "xlds tests exit $?"; tail -3 $OUT/xlds_tests.log
for lib in hip xlds hip xlds; do echo "== ksweep $lib"; PRIMX_LIB=$PWD/$CS/libprimx_$lib.so timeout 200 python tools/gemm_ksweep.py 2>&1 | grep -E "K= *(128|1152|4608)"; done | tee $OUT/ksweep.txt
for lib in hip xlds; do PRIMX_LIB=$PWD/$CS/libprimx_$lib.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events > $OUT/bench_$lib.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_$lib.json')); print('step $lib', d['repeats_ms_per_step'])"; done
timeout 200 python tools/probe/k_stride_probe.py 2>&1 | grep "K=" | tee $OUT/k_stride.txt
timeout 300 python tools/blas_ref.py 2>&1 | grep torch | tee $OUT/blas_ref.txt
# clock + MFMA utilisation (own pass: counters only with --kernel-trace)
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT -o mfma -- python bench.py --no-cpu-baseline --no-parity --no-decode-leg --steps 3 --warmup 1 --repeats 1 --no-kernel-events > /dev/null 2> $OUT/mfma.err
f=$(find $OUT -name "mfma_counter_collection.csv" | head -1); [ -n "$f" ] && (cd tools && python pmc_mfma_util.py ../$f - ../$OUT/mfma_util.txt | cut -c1-170)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --no-cpu-baseline --no-parity --no-decode-leg --steps 25 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
for db in $(find $OUT -name "*.db"); do python tools/rocprof_summary.py $db ${db%.db}_summary.txt > /dev/null; done
f=$(find $OUT -name "trace_results_summary.txt" | head -1); [ -n "$f" ] && head -12 $f | cut -c1-150
find $OUT -name "*.csv" -size +20M -delete; find $OUT -name "*.db" -size +30M -delete
du -sh $OUT
