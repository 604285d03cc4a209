#!/usr/bin/env python
"""Headline benchmark: DiT denoise-steps/s (N_prim = 2048) on the HIP path.

    python bench.py --gpus N --steps K --warmup W

A "step" is one DDIM iteration of the hot path on one batch of synthetic input: ``forward_with_cfg``
(effective batch 2B: cond + uncond) through the 28-block PrimX DiT-XL plus the fused diffusion update.
Workload at N = 1: BASELINE.json configs[1] - DiT-XL (d=1152, 28 blocks, 16 heads x 72), N_prim=2048,
1370 x 768 conditioning tokens, fp16, CFG 6, batch 1, ddim25 schedule.  N > 1: one process per GPU
(torch.distributed / RCCL), rank 0's random-init weights are broadcast once as one flat buffer, then
every rank runs its own batch with no collective inside the loop (weak scaling).

Prints ONE JSON line on rank 0: the contract fields + ``roofline`` (dominant kernel, algorithmic
FLOPs / HIP-event launch time vs the 2.5 PFLOP/s dense fp16 MFMA peak) + ``cpu_baseline`` (the CPU
oracle - a port of the reference algorithm - timed on this host on a bounded sample of the same step).
Inputs are resident in HBM before the timed region; random-init weights with the zero-initialised
adaLN / final layers overwritten (SURVEY.md section 7 "vacuous-parity trap").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

XL = dict(seq_length=2048, in_channels=68, condition_channels=768, hidden_size=1152, depth=28, num_heads=16,
          attn_proj_bias=True, cond_drop_prob=0.1, gradient_checkpointing=False)  # configs/inference_dit.yml:52-62
L_COND = 1370
PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA, MI355X_MICROARCH.md chip table


def forward_flops(N: int, L: int, D: int = 1152, depth: int = 28, Dc: int = 768, C: int = 68) -> float:
    """Algorithmic FLOPs of one DiT forward of one sample (3.1714 TF at N=2048)."""
    blk = (2 * N * D * D + 4 * L * Dc * D + 4 * N * L * D + 2 * N * D * D) + (6 * N * D * D + 4 * N * N * D + 2 * N * D * D) \
        + 16 * N * D * D + 18 * D * D
    return depth * blk + 2 * N * C * D + 2 * 256 * D + 2 * D * D + 4 * D * D + 4 * N * D * C


def random_init_(model: torch.nn.Module, seed: int) -> None:
    """Non-zero random weights drawn on the device (activations O(1) through depth, small non-zero gates)."""
    g = torch.Generator(device=next(model.parameters()).device).manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name == "null_cond_embedding":
                std = 1.0
            elif name.endswith("bias"):
                std = 0.05
            elif "adaLN_modulation" in name:
                std = 0.6 / p.shape[1] ** 0.5
            else:
                std = 1.0 / p.shape[1] ** 0.5
            p.copy_(torch.randn(p.shape, generator=g, device=p.device) * std)
    model.repack()


def step_stream(diffusion, model, x, kw):
    """Endless stream of DDIM steps: consecutive full ddim loops from the same noise."""
    while True:
        for out in diffusion.ddim_sample_loop_progressive(model.forward_with_cfg, tuple(x.shape), noise=x,
                                                          clip_denoised=False, model_kwargs=kw, device=x.device):
            yield out


def cpu_baseline(n_prim: int, budget_blocks: int = 14, threads: int = 32):
    """The CPU oracle (oracle/dit_ref.py, fp32 - the port of the reference algorithm) on a bounded
    sample: one CFG step (effective batch 2) at the full width with `budget_blocks` of the 28 blocks,
    extrapolated linearly in depth (blocks are identical in cost; embedders/final layer are < 0.1 %)."""
    from oracle import dit_ref, synth
    dit_ref.ATTN_DTYPE = torch.float32
    cfg = dict(in_channels=68, condition_channels=768, hidden_size=1152, depth=budget_blocks)
    sd = synth.dit_state_dict(0, **cfg)
    x = synth.tensor(0, "x", (1, n_prim, 68))
    y = synth.tensor(0, "y", (1, L_COND, 768))
    t = torch.tensor([960])
    torch.set_num_threads(min(threads, os.cpu_count() or 1))   # 32 is the measured optimum on the 256-core GPU-box host
    with torch.no_grad():
        t0 = time.perf_counter()
        dit_ref.dit_forward_with_cfg(sd, x, t, y, 16, 6.0)
        dt = time.perf_counter() - t0
    per_step = dt * 28.0 / budget_blocks
    return {"value": 1.0 / per_step, "unit": "denoise-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 CFG step (eff. batch 2), N_prim={n_prim}, L=1370, d=1152, {budget_blocks}/28 blocks timed "
                      f"({dt:.1f} s) and scaled x{28 // budget_blocks}; fp32 torch-CPU oracle"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="user samples per GPU (effective batch is 2x with CFG)")
    ap.add_argument("--n-prim", type=int, default=2048)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-launch HIP events (roofline leg)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    import topia_xl_amd as pkg
    from topia_xl_amd import ops
    from topia_xl_amd.sharding import broadcast_module_

    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    with torch.device(dev):                                    # construct on the GPU: no 3.6 GB host init + H2D
        model = pkg.DiT(**XL).eval()
    if rank == 0:
        random_init_(model, 42)
    wbytes = broadcast_module_(model, 0)                      # RCCL broadcast, one flat buffer (N > 1)
    B, N = args.batch, args.n_prim
    gen = torch.Generator().manual_seed(42 + rank)
    x = torch.randn(B, N, 68, generator=gen).to(dev)           # CPU draw then H2D, as inference.py:316
    y = torch.randn(B, L_COND, 768, generator=gen).to(dev)
    diffusion = pkg.create_diffusion("ddim25", noise_schedule="squaredcos_cap_v2", parameterization="v")
    kw = dict(y=y, cfg_scale=6.0, precision_dtype=dt, enable_amp=True)
    stream = step_stream(diffusion, model, x, kw)

    for _ in range(args.warmup):
        next(stream)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = next(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # Roofline leg: HIP events around every GEMM / attention launch (ops._timed, on the launch stream) over a SECOND pass of
    # the same K steps, right after the timed one (same process, buffers and clocks).  Bracketing every launch inside the
    # timed region itself was measured to cost 12.6 % (11.69 vs 10.21 ms/step at configs[1]: ~400 event packets per step,
    # each a bubble between two dependent kernels), which would deflate `value`; kernel durations are unaffected.
    prof, t_instr = None, None
    if not args.no_kernel_events and rank == 0:
        ops.PROFILE = []
        t1 = time.perf_counter()
        for _ in range(args.steps):
            next(stream)
        torch.cuda.synchronize()
        t_instr = time.perf_counter() - t1
        prof, ops.PROFILE = ops.PROFILE, None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(last["sample"]).all(), "non-finite sample"

    if rank == 0:
        steps_per_s = world * B * args.steps / elapsed
        flops_step = 2 * B * forward_flops(N, L_COND)
        res = {
            "metric": "DiT denoise-steps/sec (N_prim=2048) + samples/sec @25-step DDIM",
            "value": steps_per_s, "unit": "denoise-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16" if dt == torch.float16 else "bf16",
            "data": "synthetic (randn latents + randn conditioning tokens, random-init weights, all layers non-zero)",
            "config": {"workload": f"BASELINE configs[1]: DiT-XL d=1152 depth=28 heads=16x72, N_prim={N}, "
                                   f"L_cond={L_COND}x768, CFG 6 (eff. batch {2 * B}/GPU), batch {B}/GPU, ddim25",
                       "parallelism": f"batch-sharded replicas x{world}, no collective in the loop",
                       "weight_broadcast_bytes": wbytes},
            "samples_per_s_at_25_steps": steps_per_s / 25.0,
            "algorithmic_tflops_per_step": flops_step / 1e12,
            "achieved_tflops_whole_step": world * flops_step * args.steps / elapsed / 1e12,
            "frac_of_mfma_peak_whole_step": flops_step * args.steps / elapsed / 1e12 / PEAK_TFLOPS,
        }
        if prof:
            agg = {}
            for tag, fl, s, e in prof:
                a = agg.setdefault(tag, [0.0, 0.0, 0])
                a[0] += s.elapsed_time(e)
                a[1] += fl
                a[2] += 1
            dom = max(agg, key=lambda k: agg[k][0])
            ms, fl, n = agg[dom]
            ach = fl / (ms * 1e-3) / 1e12
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "r1_traffic.json")   # PMC pass (tools/pmc_traffic.py), per launch
            if os.path.exists(tfile):
                traffic = json.load(open(tfile)).get(dom, {}).get("hbm_bytes_per_launch")
            res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS, "traffic": traffic,
                               "launches": n, "avg_launch_ms": ms / n, "algorithmic_gflop_per_launch": fl / n / 1e9,
                               "instrumented_ms_per_step": 1e3 * t_instr / args.steps,
                               "note": "kernel name as printed by rocprofv3; HIP events bracket every launch on the launch "
                                       "stream over a second pass of the same K steps right after the timed one (inside "
                                       "the timed region they cost 12.6 % and would deflate `value`); traffic = "
                                       "(2*FETCH_SIZE + WRITE_SIZE) KiB per launch from separate --pmc passes "
                                       "(gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section)"}
            res["kernels"] = {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[2] / args.steps,
                                  "tflops": v[1] / (v[0] * 1e-3) / 1e12} for k, v in sorted(agg.items())}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(N)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
