"""Per-(kernel, shape) MFMA utilisation from ONE rocprofv3 pass of bench.py:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU \\
              GRBM_GUI_ACTIVE --output-format csv -d DIR -o NAME -- python bench.py ...
    python tools/pmc_mfma_util.py DIR/**/NAME_counter_collection.csv <DIR/**/NAME_kernel_trace.csv | -> out.txt

Columns: `busy/SIMD` = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs (cycles the matrix pipe of an average SIMD was busy);
`util@2.4` = busy/SIMD / (duration x 2.4 GHz): the utilisation against the SPEC clock - a lower bound, because the chip does
not hold 2.4 GHz under these kernels (the GEMM timelines, PRIMX_GEMM_PROF=1, measure 1.8 - 2.0 GHz as core cycles per 100 MHz
tick inside the workgroups); `util@GUI` = busy/SIMD / GRBM_GUI_ACTIVE - GUI_ACTIVE also counts the dispatch overhead of a
profiled launch, so it is only meaningful for kernels of >= 40 us.  "of 2.5PF" = algorithmic FLOPs / duration of THIS
(profiled) pass / the 2.5 PFLOP/s dense peak.  `GHz` = GRBM_GUI_ACTIVE / duration: the shader clock while the kernel ran (same caveat).
A fourth argument selects the shape tables: ddim (default), b8 (batch 8: T = 32768), c5 (configs[4] per GPU: bf16, N_prim = 4096, batch 4)."""
import collections
import csv
import re
import sys

sys.argv += [None] * 5
cfile, tfile, out, mode = sys.argv[1], sys.argv[2], sys.argv[3], (sys.argv[4] or "ddim")
import pmc_traffic  # noqa: E402
from pmc_traffic import short, tag_of   # the launch-order -> shape tables  # noqa: E402

def gemm_flops(shape):      # "MxNxK" -> 2 M N K; "PxNqxNkvxdh" -> 4 P Nq Nkv dh
    try:
        d = [int(x) for x in shape.split("x")]
    except ValueError:
        return None
    return 2 * d[0] * d[1] * d[2] if len(d) == 3 else 4 * d[0] * d[1] * d[2] * d[3] if len(d) == 4 else None


FLOPS = {"4096x1152x1152": 2 * 4096 * 1152 * 1152, "4096x1152x4608": 2 * 4096 * 1152 * 4608, "4096x4608x1152": 2 * 4096 * 4608 * 1152,
         "4096x3456x1152": 2 * 4096 * 3456 * 1152, "1536x64512x768": 2 * 1370 * 64512 * 768,
         "4096x3456x1152+1536x2304x768": 2 * 4096 * 3456 * 1152 + 2 * 1370 * 2304 * 768, "1536x2304x768": 2 * 1370 * 2304 * 768,
         "32x2048x2048x72": 4 * 32 * 2048 * 2048 * 72, "32x2048x1370x72": 4 * 32 * 2048 * 1370 * 72,
         "256->256 @4^3 x2048": 2 * 2048 * 64 * 256 * 27 * 256, "gn+256->32+sc @8^3 x2048": 2 * 2048 * 512 * 32 * 28 * 256,
         "256->32 @8^3 x2048": 2 * 2048 * 512 * 32 * 27 * 256}


dur = {}
if tfile and tfile != "-":
    for r in csv.DictReader(open(tfile)):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
rows = collections.defaultdict(dict)
meta = {}
for r in csv.DictReader(open(cfile)):
    d = int(r["Dispatch_Id"])
    rows[d][r["Counter_Name"]] = float(r["Counter_Value"])
    meta[d] = (r["Kernel_Name"], int(r.get("Grid_Size", 0)))
    if d not in dur and r.get("End_Timestamp"):      # the counter csv carries the dispatch's own timestamps too
        dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
pmc_traffic.FOLDED = any(short(m[0]).startswith("gemm144l_dma_kernel<1, 6>") for m in meta.values())
seen = collections.Counter()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(rows):
    tag = tag_of(meta[d][0], meta[d][1], seen, mode)
    for c, v in rows[d].items():
        agg[tag][c].append(v)
    if d in dur:
        agg[tag]["us"].append(dur[d])
mean = lambda v: sum(v) / len(v) if v else float("nan")
lines = ["# per (kernel, shape): means over the launches of one profiled bench pass (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE)",
         f"{'launches':>8} {'us':>8} {'busy/SIMD':>10} {'util@2.4':>8} {'util@GUI':>8} {'GHz':>5} {'MFMA insts':>11} {'VALU act/wave-cyc':>17} {'TF/s':>7} {'of 2.5PF':>8}  kernel"]
for tag in sorted(agg, key=lambda t: -sum(agg[t]["us"])):
    a = agg[tag]
    if not any(s in tag for s in ("gemm", "attn", "conv", "ln_modulate", "gemv", "groupnorm")):
        continue
    us, gui, busy = mean(a["us"]), mean(a["GRBM_GUI_ACTIVE"]), mean(a["SQ_VALU_MFMA_BUSY_CYCLES"])
    clk = gui / us / 1e3 if us else float("nan")
    div = 8.0 if clk > 4.0 else 1.0        # some rocprofv3 builds sum GUI_ACTIVE over the 8 XCDs
    gui /= div
    clk /= div
    util = busy / (1024.0 * gui) if gui else float("nan")
    shape = tag.split("> ")[-1] if "> " in tag else ""
    fl = FLOPS.get(shape) or gemm_flops(shape)
    if "convt_" in tag:                      # the k2s2 upsample shares the convolution's shape tag: 8 taps, not 27
        fl = 2 * 2048 * 64 * 8 * 256 * 256
    tf = fl / us / 1e6 if fl and us else float("nan")
    valu = mean(a["SQ_ACTIVE_INST_VALU"]) / mean(a["SQ_WAVE_CYCLES"]) if a["SQ_WAVE_CYCLES"] else float("nan")
    u24 = busy / 1024.0 / (us * 2400.0) if us else float("nan")
    lines.append(f"{len(a['us']):8d} {us:8.2f} {busy / 1024.0:10.0f} {u24:8.3f} {util:8.3f} {clk:5.2f} {mean(a['SQ_INSTS_MFMA']):11.4g} {valu:17.3f} {tf:7.0f} {tf / 2500.0:8.3f}  {tag}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
