"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output) of `bench.py` into per-kernel HBM traffic per
launch, keyed like bench.py's per-shape kernel tags ("<kernel name as rocprofv3 prints it> <shape>").  gfx950 correction
(MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled; both counters are in KiB.

    python tools/pmc_traffic.py <fetch_csv> <write_csv> <out.json> [ddim|decode|b8]

Kernels launched with several shapes are split by their position in the (fixed) launch sequence of one DiT block:
gemm144l_dma_kernel<dt, 1> runs cproj, proj (K = 1152) and fc2 (K = 4608) in that order; attn_kernel alternates
cross / self attention."""
import collections, csv, json, re, sys

CYCLES = {  # kernel name prefix -> shape tags in launch order within one block (BASELINE configs[1], fp16)
    "gemm144_dma_kernel<1, 1, 0>": ["4096x1152x1152", "4096x1152x1152", "4096x1152x4608"],   # PRIMX_GEMM_LOADER=0
    "gemm144_dma_kernel<1, 2, 0>": ["4096x1152x1152"],
    "gemm144l_dma_kernel<1, 1>": ["4096x1152x1152", "4096x1152x1152", "4096x1152x4608"],
    "gemm144l_dma_kernel<1, 2>": ["4096x1152x1152"],
    "gemm288q_dma_kernel<1, 0>": ["4096x4608x1152"],
    "gemm288q_dma_kernel<1, 0, 64>": ["4096x4608x1152"],
    "gemm288p_dma_kernel<1, false>": ["4096x4608x1152"],     # (rounds 3 - 5 traces)
    # with the LayerNorm fold (DiT.fold_ln, the default in planned loops): producers <1, 6> = cproj, proj, fc2 of every block but the
    # last one's fc2, which stays <1, 1>; consumers <1, 7> (to_q from block 1 on; qkv on the 256 x 288 tile), fc1 on gemm288q<1, 8, 64>
    "gemm144l_dma_kernel<1, 6>": ["4096x1152x1152", "4096x1152x1152", "4096x1152x4608"] * 27 + ["4096x1152x1152"] * 2,
    "gemm144l_dma_kernel<1, 7>": ["4096x1152x1152"],
    "gemm288q_dma_kernel<1, 7>": ["4096x3456x1152"],
    "gemm288q_dma_kernel<1, 7, 32>": ["4096x3456x1152"],     # (rocprofv3 prints the ring's template argument since round 6)
    "gemm288q_dma_kernel<1, 7, 64>": ["4096x3456x1152"],     # (the heads epilogues on the 128-byte ring: end of round 6)
    "gemm288q_dma_kernel<1, 8, 64>": ["4096x4608x1152"],
    "gemm288q_pair_kernel<1>": ["4096x3456x1152+1536x2304x768"],
    "gemm288q_pair_kernel<1, 64>": ["4096x3456x1152+1536x2304x768"],
    "gemm288q_pair_kernel<1, 32>": ["4096x3456x1152+1536x2304x768"],   # qkv + the next block's to_k / to_v riding on its idle CUs (ABI 25)
    "gemm288p_dma_kernel<1, true>": ["4096x4608x1152"],      # (rounds 4 - 5 traces)
    "attn_kernel<1, 5, 3, 0, 0>": ["32x2048x1370x72", "32x2048x2048x72"],
    # --config decode (2048 primitives): one shape per kernel
    "conv3_s4c256_kernel<1, 0>": ["256->256 @4^3 x2048"],
    "conv3_s8c256n32_kernel<1, 0>": ["256->32 @8^3 x2048"],
    "conv3_s8c256n32_kernel<1, 1>": ["gn+256->32+sc @8^3 x2048"],
    "convt_s4c256_kernel<1>": ["256->256 @4^3 x2048"],
}


# BASELINE configs[2] / [3] per-GPU shape (batch 8, T = 32768 token rows, fp16): every GEMM on the 256 x 288 tile.  The kernels that run several
# shapes are split by launch order (producers <1, 6, 64>: cproj, proj, fc2 per block) or by grid size (heads consumers <1, 7, 32>: qkv 1536
# workgroups, to_q 512; <1, 2, 32>: the batched K / V projection, block 0's to_q)
CYCLES_B8 = {
    "gemm288q_dma_kernel<1, 6, 64>": ["32768x1152x1152", "32768x1152x1152", "32768x1152x4608"] * 27 + ["32768x1152x1152"] * 2,
    "gemm288q_dma_kernel<1, 1, 64>": ["32768x1152x4608"],
    "gemm288q_dma_kernel<1, 8, 64>": ["32768x4608x1152"],
    "attn_kernel<1, 5, 3, 0, 0>": ["256x2048x1370x72", "256x2048x2048x72"],
}
GRID_B8 = {   # kernel -> {workgroups: shape}
    "gemm288q_dma_kernel<1, 7, 32>": {1536: "32768x3456x1152", 512: "32768x1152x1152"},
    "gemm288q_dma_kernel<1, 2, 32>": {512: "32768x1152x1152", 10752: "12288x64512x768", 21504: "24576x64512x768"},
    "gemm288q_dma_kernel<1, 7, 64>": {1536: "32768x3456x1152", 512: "32768x1152x1152"},
    "gemm288q_dma_kernel<1, 2, 64>": {512: "32768x1152x1152", 10752: "12288x64512x768", 21504: "24576x64512x768"},
}


def tag_of(kname, grid, seen, mode="ddim"):
    """(kernel, shape) tag of one dispatch: `grid` = Grid_Size (threads), `seen` = a Counter of the kernel's earlier dispatches."""
    k = short(kname)
    if mode == "c5":   # configs[4] per GPU: bf16, N_prim = 4096, batch 4 (effective 8): only the attention launches are tagged
        if k.startswith("attn_kernel<2,"):
            t = f"{k} {['128x4096x1370x72', '128x4096x4096x72'][seen[k] % 2]}"
            seen[k] += 1
            return t
        return k
    if mode == "b8":
        if k in GRID_B8:
            return f"{k} {GRID_B8[k].get(grid // 512, str(grid // 512) + ' workgroups')}"
        cyc = CYCLES_B8.get(k)
    else:
        if k.startswith("gemm288q_dma_kernel<1, 2>") or k.startswith("gemm288q_dma_kernel<1, 2, 32>") or k.startswith("gemm288q_dma_kernel<1, 2, 64>"):
            # (the batched K / V projection; block 0's projection alone on the riders' tile kernel - 48 workgroups, ABI 25; block 0's to_q)
            return k + (" 1536x64512x768" if grid > 512 * 400 else " 1536x2304x768" if grid == 512 * 48 else " 4096x3456x1152")
        cyc = CYCLES.get(k)
        if cyc and k == "gemm144l_dma_kernel<1, 1>" and FOLDED:
            cyc = ["4096x1152x4608"]                       # only the last block's fc2 is left on the plain gate-residual kernel
    if cyc:
        t = f"{k} {cyc[seen[k] % len(cyc)]}"
        seen[k] += 1
        return t
    return k


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


FOLDED = False   # set by load(): the trace contains fold kernels
MODE = "ddim"    # "b8": the batch-8 shapes (argv[4])


def load(path, counter):
    global FOLDED
    acc = collections.defaultdict(list)
    seen = collections.Counter()
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    FOLDED = any(short(r["Kernel_Name"]).startswith("gemm144l_dma_kernel<1, 6>") for r in rows)
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    for r in rows:
        acc[tag_of(r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0), seen, MODE)].append(float(r["Counter_Value"]))
    return acc


def main():
    global MODE
    MODE = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] else "ddim"
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        if not any(s in k for s in ("gemm", "attn", "ln_modulate", "gemv", "conv", "groupnorm")):
            continue
        fk = sum(f.get(k, [0])) / max(len(f.get(k, [])), 1)
        wk = sum(w.get(k, [0])) / max(len(w.get(k, [])), 1)
        out[k] = {"launches_sampled": len(f.get(k, [])), "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
                  "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        print(f"{v['hbm_bytes_per_launch']/1e6:10.1f} MB/launch  fetch {v['FETCH_SIZE_KiB']:12.0f} KiB  write {v['WRITE_SIZE_KiB']:12.0f} KiB  n={v['launches_sampled']:5d}  {k}")


if __name__ == "__main__":
    main()
