#!/usr/bin/env python
"""Headline benchmark of the hot path on the HIP device.

    python bench.py --gpus N --steps K --warmup W [--config ddim|decode|c4] [--batch B] [--repeats R]

--config ddim (default): metric = DiT denoise-steps/s.  A "step" is one DDIM iteration on one batch of synthetic input:
  `forward_with_cfg` (effective batch 2B: cond + uncond) through the 28-block PrimX DiT-XL plus the fused diffusion
  update.  Workload at N = 1: BASELINE.json configs[1] - DiT-XL (d=1152, 28 blocks, 16 heads x 72), N_prim=2048,
  1370 x 768 conditioning tokens, fp16, CFG 6, batch 1, ddim25 schedule.
--config decode: SURVEY.md section 8d metric (2)'s second leg - a "step" is `latents_to_primitives` of one batch of
  samples (latent de-normalisation + VAE.decode of B x 2048 primitives + inverse normalisation); metric = samples/s.
--config c4: BASELINE.json configs[3] - batch 8, 100 DDIM steps + decode: K DDIM steps at batch 8 are timed and the
  decode is timed separately; samples/s = 8 / (100 x step + decode).

N > 1: one process per GPU (torch.distributed / RCCL); rank 0's packed 16-bit weight blob is broadcast once, then every
rank runs its own batch with no collective inside the loop (weak scaling).  The timed region is K steps between
barrier + synchronize, repeated R times (median reported); MAX over ranks.  Started WITHOUT a
launcher (`python bench.py --gpus N`, no WORLD_SIZE in the environment) the script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`; every rank checks that the
process group really has N ranks and that its GPU exists, and the line reports `n_gpus: N`, per-rank min / max
ms/step and `weight_broadcast_bytes`.  `--dry-run` stops after the rendezvous (gloo when there is no GPU: the launcher's
CPU test).

The default (ddim) line also carries the rest of the metric OUTSIDE the headline's timed region: `decode` (the VAE leg
on this GPU: ms, samples/s, its own roofline / cpu_baseline / parity) and `samples_per_s_measured` (a whole 25-step
DDIM loop + decode of the sample, timed end to end, median of R) - `value` stays the denoise-steps/s of K steps.

Prints ONE JSON line on rank 0: the contract fields + `roofline` (dominant kernel BY SHAPE, algorithmic FLOPs /
HIP-event launch time vs the 2.5 PFLOP/s dense fp16 MFMA peak) + `cpu_baseline` (the CPU oracle - a port of the
reference algorithm - timed on this host on a bounded sample of the same step) + `parity` (the benchmarked model's own
output against the fp32 golden of the REAL reference at the full 28-block configuration, and against the fp32 oracle
forward the cpu_baseline leg computes).  Weights: deterministic synthetic (oracle/synth.py, every layer non-zero - the
"vacuous-parity trap" of SURVEY.md section 7); inputs resident in HBM before the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

XL = dict(seq_length=2048, in_channels=68, condition_channels=768, hidden_size=1152, depth=28, num_heads=16,
          attn_proj_bias=True, cond_drop_prob=0.1, gradient_checkpointing=False)  # configs/inference_dit.yml:52-62
VAE_CFG = dict(in_channels=6, latent_channels=1, out_channels=6, down_channels=[32, 256], mid_attention=True,
               up_channels=[256, 32], layers_per_block=2, gradient_checkpointing=False)  # configs/inference_dit.yml:32-43
L_COND = 1370
PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA, MI355X_MICROARCH.md chip table
VAE_FLOPS_PER_PRIM = 2.2428e9
WEIGHT_SEED = 4321     # = tests/golden/make_golden_xl.py XL_SEED: the benchmarked weights are the golden's weights


def forward_flops(N: int, L: int, D: int = 1152, depth: int = 28, Dc: int = 768, C: int = 68) -> float:
    """Algorithmic FLOPs of one DiT forward of one sample (3.1714 TF at N=2048)."""
    blk = (2 * N * D * D + 4 * L * Dc * D + 4 * N * L * D + 2 * N * D * D) + (6 * N * D * D + 4 * N * N * D + 2 * N * D * D) \
        + 16 * N * D * D + 18 * D * D
    return depth * blk + 2 * N * C * D + 2 * 256 * D + 2 * D * D + 4 * D * D + 4 * N * D * C


def kv_projection_flops(L: int, D: int = 1152, depth: int = 28, Dc: int = 768) -> float:
    """to_k + to_v of every block for one sample (the step-invariant part, counted in forward_flops)."""
    return depth * 4.0 * L * Dc * D


def load_synth_weights(model: torch.nn.Module, depth: int) -> None:
    from oracle import synth   # deterministic generator only (no oracle arithmetic): same tensors as the goldens
    sd = synth.dit_state_dict(WEIGHT_SEED, in_channels=68, condition_channels=768, hidden_size=1152, depth=depth)
    model.load_state_dict(sd, strict=True)


def step_stream(diffusion, model, x, kw):
    """Endless stream of DDIM steps: consecutive full ddim loops from the same noise."""
    while True:
        for out in diffusion.ddim_sample_loop_progressive(model.forward_with_cfg, tuple(x.shape), noise=x,
                                                          clip_denoised=False, model_kwargs=kw, device=x.device):
            yield out


def cpu_baseline_ddim(n_prim: int, budget_blocks: int = 28, threads: int = 32):
    """The CPU oracle (oracle/dit_ref.py, fp32 - the port of the reference algorithm, NOT the reference's own modules: /root/reference
    does not exist on the GPU box) on a bounded sample: ONE whole CFG step (effective batch 2) of the full model - all 28 blocks, ~20 s
    on 32 host threads (`budget_blocks` < 28: that many blocks, scaled linearly in depth).  Returns (record, oracle output, inputs) -
    the output is what the `parity` leg compares the GPU's forward of the SAME model with."""
    from oracle import dit_ref, synth
    dit_ref.ATTN_DTYPE = torch.float32
    sd = synth.dit_state_dict(WEIGHT_SEED, in_channels=68, condition_channels=768, hidden_size=1152, depth=budget_blocks)
    x = synth.tensor(WEIGHT_SEED, "xl_c2.x", (1, n_prim, 68))
    y = synth.tensor(WEIGHT_SEED, "xl_c2.y", (1, L_COND, 768))
    t = torch.tensor([800])
    torch.set_num_threads(min(threads, os.cpu_count() or 1))   # 32 is the measured optimum on the 256-core GPU-box host
    with torch.no_grad():
        t0 = time.perf_counter()
        out = dit_ref.dit_forward_with_cfg(sd, x, t, y, 16, 6.0)
        dt = time.perf_counter() - t0
    per_step = dt * 28.0 / budget_blocks
    rec = {"value": 1.0 / per_step, "unit": "denoise-steps/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"1 CFG step (eff. batch 2), N_prim={n_prim}, L=1370, d=1152, {budget_blocks}/28 blocks timed "
                     f"({dt:.1f} s)" + ("" if budget_blocks == 28 else f" and scaled x{28.0 / budget_blocks:g}") +
                     "; fp32 torch-CPU port of the reference algorithm (oracle/dit_ref.py), not the reference's own modules"}
    return rec, out, (sd, x, y, t, budget_blocks)


def cpu_baseline_decode(vae_sd, n_prims: int = 96, threads: int = 32):
    """oracle/vae_ref.py (fp32 port of VAE.decode) on a bounded sample of primitives; samples/s = prims/s / 2048."""
    from oracle import synth, vae_ref
    z = synth.tensor(WEIGHT_SEED, "bench.vae.z", (n_prims, 1, 4, 4, 4))
    torch.set_num_threads(min(threads, os.cpu_count() or 1))
    with torch.no_grad():
        vae_ref.vae_decode(vae_sd, z[:8])
        t0 = time.perf_counter()
        out = vae_ref.vae_decode(vae_sd, z)
        dt = time.perf_counter() - t0
    rec = {"value": n_prims / dt / 2048.0, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"VAE.decode of {n_prims} primitives ({dt:.1f} s), scaled to 2048 primitives per sample; fp32 torch-CPU oracle"}
    return rec, out, z


def torch_rocm_reference(model, x, y, dt, dev, batches=(1, 8), iters: int = 4):
    """The reference's forward_with_cfg written with STOCK PyTorch-ROCm ops on this GPU, outside every timed region of the
    product: oracle/dit_ref.py (the op-by-op port of models/dit_crossattn.py) run on the device under torch.autocast - F.linear
    -> hipBLASLt, F.layer_norm, F.gelu, and F.scaled_dot_product_attention where the reference calls xformers'
    memory_efficient_attention (not installable here).  Same weights as the benchmarked model, pre-cast to the 16-bit type once (the
    best case for autocast, which would re-cast them every forward).  One forward_with_cfg per "step" (the sampler update is
    < 0.2 % of a step).  A baseline on the same chip, next to `cpu_baseline`: not the product and not the target."""
    import torch.nn.functional as F
    from oracle import dit_ref

    def sdpa(q, k, v, scale):                                      # [B, M, H, K] as xformers takes them
        o = F.scaled_dot_product_attention(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), scale=scale)
        return o.permute(0, 2, 1, 3)

    rec = {"what": "oracle/dit_ref.py on the device under torch.autocast: stock F.linear (hipBLASLt) / F.scaled_dot_product_attention / "
                   "F.layer_norm, weights pre-cast to the 16-bit type; one forward_with_cfg per step, HIP events, median",
           "torch": torch.__version__}
    keep = dit_ref.attention_core
    try:
        sd = {k: (v.detach().to(dt) if v.is_floating_point() else v.detach()) for k, v in model.state_dict().items()}
        dit_ref.attention_core = sdpa
        g = torch.Generator().manual_seed(78)
        for bs in batches:
            xs = x if bs == x.shape[0] else torch.randn(bs, x.shape[1], x.shape[2], generator=g).to(dev)
            ys = y if bs == y.shape[0] else torch.randn(bs, y.shape[1], y.shape[2], generator=g).to(dev)
            tt = torch.full((bs,), 800, dtype=torch.int64, device=dev)
            ms = []
            with torch.no_grad(), torch.device(dev), torch.autocast("cuda", dtype=dt):
                for i in range(iters + 2):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    out = dit_ref.dit_forward_with_cfg(sd, xs, tt, ys, 16, 6.0)
                    e.record()
                    torch.cuda.synchronize()
                    if i >= 2:
                        ms.append(s.elapsed_time(e))
            assert torch.isfinite(out.float()).all()
            m = statistics.median(ms)
            rec[f"batch{bs}"] = {"ms_per_step": m, "value": bs * 1e3 / m, "unit": "denoise-steps/s"}
            del out
        del sd
    except Exception as ex:  # a stock op this torch build lacks on gfx950 must not take the bench line down
        rec["error"] = f"{type(ex).__name__}: {ex}"[:300]
    finally:
        dit_ref.attention_core = keep
        torch.cuda.empty_cache()
    return rec


def timed_repeats(run_steps, steps: int, repeats: int, world: int, dist, dev):
    """R x (barrier, synchronize, K steps, synchronize, barrier); per-repeat elapsed = MAX over ranks."""
    out = []
    mine = []
    for _ in range(repeats):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(steps)
        torch.cuda.synchronize()
        local = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
            tm = torch.tensor([local], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            out.append(float(tm.item()))
        else:
            out.append(local)
        mine.append(local)
    return out, mine


def traffic_file() -> str:
    """The newest committed PMC traffic table (profiles/r<round>_traffic.json)."""
    import glob
    import re
    pat = re.compile(r"^r(\d+)_traffic\.json$")       # (r6_traffic.json; not r6_traffic_b8.json, not an archived r6a_traffic.json)
    fs = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")) if pat.match(os.path.basename(f))),
                key=lambda f: int(pat.match(os.path.basename(f)).group(1)))
    return fs[-1] if fs else os.path.join(ROOT, "profiles", "none")


def event_pair_overhead_ms(n: int = 200) -> float:
    """What a (start, end) pair of HIP events reads with NOTHING between them on the launch stream: the part of every per-launch
    reading that is not the kernel.  Measured in the same process right after the per-launch pass (median of n pairs)."""
    vals = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        e.record()
        vals.append((s, e))
    torch.cuda.synchronize()
    return statistics.median(s.elapsed_time(e) for s, e in vals)


def kernel_report(prof, steps: int, traffic_file: str, overhead_ms: float = 0.0):
    """Per-kernel times of the per-launch pass.  Every reading has HALF the measured empty event-pair reading subtracted: the raw
    readings of a step's launches summed to more than the event-free step (round-4 review: + 4 - 5 %), and a reading minus the
    whole empty-pair reading falls BELOW the same launch's rocprofv3 duration (round 5: fc1 49.8 us against 51.9 - 54.6) - the
    packet processor overlaps about half of a marker's latency with the kernel behind it.  With the half the per-launch
    averages agree with the rocprofv3 trace of the same command (profiles/README.md, r5) and the sum stays below the step."""
    agg = {}
    for tag, fl, s, e in prof:
        a = agg.setdefault(tag, [0.0, 0.0, 0, 0.0])
        raw = s.elapsed_time(e)
        a[0] += raw - min(0.5 * overhead_ms, 0.5 * raw)
        a[1] += fl
        a[2] += 1
        a[3] += raw
    mfma = {k: v for k, v in agg.items() if v[1] > 0}
    dom = max(mfma, key=lambda k: mfma[k][0])
    ms, fl, n, _raw = agg[dom]
    ach = fl / (ms * 1e-3) / 1e12
    traffic, src = None, None
    if os.path.exists(traffic_file):
        rec = json.load(open(traffic_file)).get(dom)
        if rec:
            traffic = rec.get("hbm_bytes_per_launch")
            src = f"static profile {os.path.relpath(traffic_file, ROOT)} (separate rocprofv3 --pmc passes of this command; not re-measured in this run)"
    roof = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS,
            "traffic": traffic, "traffic_source": src, "launches": n, "avg_launch_ms": ms / n,
            "algorithmic_gflop_per_launch": fl / n / 1e9, "avg_launch_ms_raw_event_reading": _raw / n,
            "event_pair_overhead_ms": overhead_ms,
            "note": "avg_launch_ms = HIP-event reading minus HALF the reading of an empty event pair measured in the same process "
                    "(event_pair_overhead_ms; the raw reading stands next to it; see kernel_report); kernel name as rocprofv3 prints it + the launch shape (GEMM MxNxK / attention problems x Nq x Nkv x dh); "
                    "HIP events bracket every launch on the launch stream over a separate pass right after the timed "
                    "one (inside the timed region they cost 12.6 % and would deflate `value`); traffic = (2*FETCH_SIZE "
                    "+ WRITE_SIZE) KiB per launch (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section)"}
    kernels = {k: {"ms_per_step": v[0] / steps, "launches_per_step": v[2] / steps,
                   "tflops": (v[1] / (v[0] * 1e-3) / 1e12) if v[1] else None} for k, v in sorted(agg.items())}
    roof["kernels_sum_ms_per_step"] = sum(v[0] for v in agg.values()) / steps
    roof["kernels_sum_ms_per_step_raw"] = sum(v[3] for v in agg.values()) / steps
    return roof, kernels


def workload_name(B: int, N: int, dtype: str, schedule: str) -> str:
    """Name the BASELINE.json configuration the (batch, N_prim, dtype) of this run corresponds to."""
    base = f"DiT-XL d=1152 depth=28 heads=16x72, N_prim={N}, L_cond={L_COND}x768, CFG 6 (eff. batch {2 * B}/GPU), batch {B}/GPU, {dtype}, {schedule}"
    if (B, N, dtype) == (1, 2048, "fp16"):
        return "BASELINE configs[1]: " + base
    if (B, N) == (8, 2048):
        return "BASELINE configs[2] per-GPU shape (batch 64 over 8 GPUs -> 8 per GPU): " + base
    if (B, N, dtype) == (4, 4096, "bf16"):
        return "BASELINE configs[4] per-GPU shape (batch 16 over 4 GPUs -> 4 per GPU): " + base
    return "not a BASELINE configuration (custom --batch / --n-prim / --dtype): " + base


def self_launch(n: int) -> None:
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def rendezvous(args):
    """(world, rank, local, dev, dist) with the process group up for world > 1; loud failures instead of a silent 1-rank run."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    have_gpu = torch.cuda.is_available()
    if not have_gpu and not args.dry_run:
        raise SystemExit("bench.py: no HIP device visible (only --dry-run works without one)")
    # PRIMX_BENCH_SHARE_GPU=1 (tests/test_hip_rccl.py only): every rank runs on cuda:0 and the process group is gloo - the N > 1 code
    # path of this file (weight broadcast into a packed-only model, all_gather of ranks and times, MAX over ranks) on a ONE-GPU box.
    # RCCL refuses two ranks on one device; the line it prints says `"test_mode"` and measures nothing.
    share = os.environ.get("PRIMX_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    if have_gpu and torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local} but only {torch.cuda.device_count()} are visible")
    dev = torch.device("cuda", local) if have_gpu else torch.device("cpu")
    if have_gpu:
        torch.cuda.set_device(dev)
    # PRIMX_FORCE_COLLECTIVES=1 (tests/test_hip_rccl.py): a ONE-rank RCCL process group is brought up too and the weight broadcast is
    # really issued on it (sharding.FORCE_COLLECTIVES) - proves the collective path on a single-GPU box, measures nothing
    if world > 1 or (os.environ.get("PRIMX_FORCE_COLLECTIVES") == "1" and "MASTER_PORT" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if have_gpu and not share:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        seen = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(seen)
        if int(seen.item()) != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: the process group has {int(seen.item())} ranks, --gpus asked for {args.gpus}")
    return world, rank, local, dev, dist


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=3, help="timed repeats of K steps; the median is reported")
    ap.add_argument("--config", default="ddim", choices=["ddim", "decode", "c4"])
    ap.add_argument("--batch", type=int, default=None, help="user samples per GPU (default 1; c4: 8)")
    ap.add_argument("--n-prim", type=int, default=2048)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--reuse-cond-kv", action="store_true",
                    help="opt-in exact algebra: compute the step-invariant cross-attention K/V once per sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-launch HIP events (roofline leg)")
    ap.add_argument("--no-decode-leg", action="store_true", help="ddim: skip the decode sub-record and the measured 25-step samples/s")
    ap.add_argument("--no-side-legs", action="store_true", help="ddim at batch 1: skip the `batch8` and `bf16` sub-records")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous only: print the ranks that met (launcher check; gloo without a GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)
    world, rank, local, dev, dist = rendezvous(args)
    if args.dry_run:
        ranks = [None] * world
        if world > 1:
            dist.all_gather_object(ranks, (rank, local, str(dev)))
            dist.destroy_process_group()
        else:
            ranks = [(rank, local, str(dev))]
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": ranks}), flush=True)
        return

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    import topia_xl_amd as pkg
    from topia_xl_amd import ops, pipeline
    from topia_xl_amd.sharding import broadcast_module_, broadcast_packed_

    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    B = args.batch if args.batch is not None else (8 if args.config == "c4" else 1)
    N = args.n_prim
    res = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f16" if dt == torch.float16 else "bf16"}

    # ------------------------------------------------------------------ models
    model = vae = None
    wbytes, wb_ms = 0, 0.0
    if args.config in ("ddim", "c4"):
        with torch.device(dev):                                    # construct on the GPU: no 3.6 GB host init + H2D
            model = pkg.DiT(**XL).eval()
        if rank == 0:
            load_synth_weights(model, 28)
        model.reuse_cond_kv = bool(args.reuse_cond_kv)
        pack_ms = 0.0
        if world > 1:                                              # rank 0's fp32 -> 16-bit pack and the receivers' allocation are
            torch.cuda.synchronize()                               # timed on their own: weight_broadcast_ms is the collective alone
            tp0 = time.perf_counter()
            model.packed(dt) if rank == 0 else model.packed_alloc(dt)
            torch.cuda.synchronize()
            pack_ms = 1e3 * (time.perf_counter() - tp0)
            dist.barrier()
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        wbytes = broadcast_packed_(model, dt, 0)                  # RCCL: the packed 16-bit blob (N > 1)
        torch.cuda.synchronize()
        wb_ms = 1e3 * (time.perf_counter() - tb0) if wbytes else 0.0
    decode_leg = args.config == "ddim" and not args.no_decode_leg
    if args.config in ("decode", "c4") or decode_leg:
        from oracle import synth
        with torch.device(dev):
            vae = pkg.VAE(**VAE_CFG).eval()
        vae_sd = synth.state_dict_like(WEIGHT_SEED, {k: v.cpu() for k, v in vae.state_dict().items()})
        if rank == 0:
            vae.load_state_dict(vae_sd, strict=True)
        wbytes += broadcast_module_(vae, 0)
        vae.compute_dtype = dt

    from oracle import synth
    gen = torch.Generator().manual_seed(42 + rank)
    if B == 1 and rank == 0 and N == 2048:
        x = synth.tensor(WEIGHT_SEED, "xl_c2.x", (1, N, 68)).to(dev)          # the golden's inputs (parity leg)
        y = synth.tensor(WEIGHT_SEED, "xl_c2.y", (1, L_COND, 768)).to(dev)
    else:
        x = torch.randn(B, N, 68, generator=gen).to(dev)           # CPU draw then H2D, as inference.py:316
        y = torch.randn(B, L_COND, 768, generator=gen).to(dev)
    mean = torch.linspace(-0.2, 0.2, 68).tolist()                  # stand-ins for configs/inference_dit.yml:64-65
    std = torch.linspace(0.8, 1.2, 68).tolist()

    # ------------------------------------------------------------------ the step
    if args.config in ("ddim", "c4"):
        diffusion = pkg.create_diffusion("ddim25" if args.config == "ddim" else "ddim100", noise_schedule="squaredcos_cap_v2",
                                         parameterization="v")
        kw = dict(y=y, cfg_scale=6.0, precision_dtype=dt, enable_amp=True)
        stream = step_stream(diffusion, model, x, kw)
        last = {}

        def run_steps(k):
            for _ in range(k):
                last["out"] = next(stream)
    else:
        samples = x

        def run_steps(k):
            for _ in range(k):
                last_dec["out"] = pipeline.latents_to_primitives(samples, vae, mean, std)
        last_dec = {}

    run_steps(args.warmup)
    elapsed_all, mine = timed_repeats(run_steps, args.steps, max(1, args.repeats), world, dist, dev)
    elapsed = statistics.median(elapsed_all)
    if args.config in ("ddim", "c4"):
        assert torch.isfinite(last["out"]["sample"]).all(), "non-finite sample"
    else:
        assert torch.isfinite(last_dec["out"]).all(), "non-finite decode"
    # what a SCALE record can be audited with: the ranks that really took part (all_gather over the process group, not the
    # launcher's word), every rank's own median ms per step, the weight broadcast's wall time
    ranks_seen = [[rank, local, torch.cuda.get_device_name(dev)]]
    ms = [1e3 * statistics.median(mine) / args.steps]
    if world > 1:
        t = torch.tensor([statistics.median(mine)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allr, t)
        ms = [1e3 * float(a.item()) / args.steps for a in allr]
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, [rank, local, torch.cuda.get_device_name(dev)])
        wbt = torch.tensor([wb_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(wbt, op=dist.ReduceOp.MAX)
        wb_ms = float(wbt.item())
    per_rank = {"min": min(ms), "max": max(ms), "all": ms}

    # Roofline leg: HIP events around every MFMA-kernel launch (ops._timed, on the launch stream) over a SEPARATE pass of the
    # same K steps, right after the timed ones (same process, buffers and clocks).
    prof = None
    if not args.no_kernel_events and rank == 0:
        ops.PROFILE = []
        run_steps(args.steps)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        ev_over = event_pair_overhead_ms()

    # Opt-in exact algebra, reported NEXT TO the headline, never as it: the same K steps with the
    # step-invariant cross-attention K / V projections computed once per conditioning tensor (DiT.reuse_cond_kv).
    reuse = None
    if args.config == "ddim" and not args.reuse_cond_kv and rank == 0 and world == 1:
        model.reuse_cond_kv = True
        run_steps(args.warmup)
        el, _ = timed_repeats(run_steps, args.steps, max(1, args.repeats), 1, dist, dev)
        model.reuse_cond_kv = False
        e = statistics.median(el)
        ex = 2 * B * forward_flops(N, L_COND) - B * kv_projection_flops(L_COND)
        reuse = {"ms_per_step": 1e3 * e / args.steps, "value": B * args.steps / e, "unit": "denoise-steps/s",
                 "executed_tflops_per_step": ex / 1e12,
                 "note": "DiT.reuse_cond_kv=True: to_k(y) / to_v(y) of all blocks computed once per conditioning tensor "
                         "(models/attention.py:106-107 do not depend on t); identical samples (tests/test_hip_fullconfig.py); "
                         "NOT the headline value - that one projects the conditioning tokens in every step"}

    # The same K steps with the unconditional half's conditioning rows projected and read EXPANDED (DiT.dedup_null_kv = False): what
    # the headline's de-duplication of `null_cond_embedding.expand_as(y)` saves, for the record.
    expanded = None
    if args.config == "ddim" and rank == 0 and world == 1 and getattr(model, "dedup_null_kv", False):
        model.dedup_null_kv = False
        run_steps(args.warmup)
        el, _ = timed_repeats(run_steps, args.steps, max(1, args.repeats), 1, dist, dev)
        model.dedup_null_kv = True
        run_steps(1)
        e = statistics.median(el)
        expanded = {"ms_per_step": 1e3 * e / args.steps, "value": B * args.steps / e, "unit": "denoise-steps/s",
                    "note": "DiT.dedup_null_kv=False (PRIMX_NULL_KV_DEDUP=0): the L identical conditioning rows of the unconditional "
                            "half are projected by to_k / to_v, stored and read like the conditional ones; bit-identical attention "
                            "results (tests/test_hip_attention.py::test_broadcast_key_value_entries)"}

    # The same K steps with the LayerNorm fold off (DiT.fold_ln = False: 85 LayerNorm launches per forward instead of 2): what the
    # fold saves, for the record (it changes rounding points, not accuracy: tests/test_hip_fold.py, tests/test_hip_fullconfig.py).
    unfolded = None
    if (args.config == "ddim" and rank == 0 and world == 1 and not args.no_side_legs and getattr(model, "fold_ln", False)
            and model._fold_ok(2 * B * N, N)):
        model.fold_ln = False
        run_steps(args.warmup)
        el, _ = timed_repeats(run_steps, args.steps, max(1, args.repeats), 1, dist, dev)
        model.fold_ln = True
        run_steps(1)
        e = statistics.median(el)
        unfolded = {"ms_per_step": 1e3 * e / args.steps, "value": B * args.steps / e, "unit": "denoise-steps/s",
                    "note": "DiT.fold_ln=False (PRIMX_DIT_FOLD=0): every LayerNorm + modulate as a launch of its own between the "
                            "gate-residual GEMM and the Linear it feeds (the round-3 / early round-4 path)"}

    # The same K steps with the conditioning K / V projection as ONE batched launch per forward (DiT.kv_ride = False) instead of riding on
    # the qkv launches' idle CUs (round 6, ABI 25): what the riders save, for the record (bit-identical samples).
    batched_kv = None
    if (args.config == "ddim" and rank == 0 and world == 1 and not args.no_side_legs and not args.reuse_cond_kv
            and getattr(model, "kv_ride", False) and getattr(model, "fold_ln", False) and model._fold_ok(2 * B * N, N)):
        model.kv_ride = False
        run_steps(args.warmup)
        el, _ = timed_repeats(run_steps, args.steps, max(1, args.repeats), 1, dist, dev)
        model.kv_ride = True
        run_steps(1)
        e = statistics.median(el)
        batched_kv = {"ms_per_step": 1e3 * e / args.steps, "value": B * args.steps / e, "unit": "denoise-steps/s",
                      "note": "DiT.kv_ride=False (PRIMX_DIT_KV_RIDE=0): to_k / to_v of the conditioning tokens of all blocks as one GEMM per "
                              "forward; the headline runs block i + 1's projection on the CUs block i's qkv launch (192 of 256 tiles) leaves "
                              "idle (primx_linear_heads_fold_pair); same tile kernel, bit-identical samples (tests/test_hip_fold.py)"}

    # Two more shapes of the same loop, reported NEXT TO the headline (outside its timed region, like `decode`): the configs[2] /
    # configs[3] per-GPU batch of 8 (T = 32768 tokens per launch: every GEMM on the 256 x 288 tile) and configs[1] in bf16 - the
    # north star's target dtype.  Same model, same kernels, K steps between synchronisations, median of R.
    side = {}
    # At N > 1 the `batch8` leg runs on EVERY rank (barrier on both sides, MAX over ranks, value = whole job) - the configs[2] shape
    # of a SCALE record; the bf16 and stock-PyTorch legs stay single-GPU context.
    if args.config == "ddim" and B == 1 and N == 2048 and not args.no_side_legs and not args.reuse_cond_kv:
        def side_leg(bs, sdt, k):
            g2 = torch.Generator().manual_seed(77 + rank)
            xs = x if bs == 1 else torch.randn(bs, N, 68, generator=g2).to(dev)
            ys = y if bs == 1 else torch.randn(bs, L_COND, 768, generator=g2).to(dev)
            st = step_stream(diffusion, model, xs, dict(y=ys, cfg_scale=6.0, precision_dtype=sdt, enable_amp=True))

            def run(kk):
                for _ in range(kk):
                    last["side"] = next(st)
            run(2)
            el, own = timed_repeats(run, k, max(1, args.repeats), world, dist, dev)
            assert torch.isfinite(last["side"]["sample"]).all(), "non-finite sample (side leg)"
            e = statistics.median(el)
            own_ms = [1e3 * statistics.median(own) / k]
            if world > 1:
                tt = torch.tensor(own_ms, dtype=torch.float64, device=dev)
                allr = [torch.zeros_like(tt) for _ in range(world)]
                dist.all_gather(allr, tt)
                own_ms = [float(a.item()) for a in allr]
            fl = 2 * bs * forward_flops(N, L_COND)
            ln = 64 + (L_COND % 64 if L_COND > 64 else 0)
            ex = fl - (bs * kv_projection_flops(L_COND) * (1.0 - ln / L_COND) if getattr(model, "dedup_null_kv", False) else 0.0)
            return {"workload": workload_name(bs, N, "fp16" if sdt == torch.float16 else "bf16", "ddim25"), "steps": k,
                    "ms_per_step": 1e3 * e / k, "value": world * bs * k / e, "unit": "denoise-steps/s",
                    "repeats_ms_per_step": [1e3 * v / k for v in el], "per_rank_ms_per_step": own_ms,
                    "algorithmic_tflops_per_step": fl / 1e12, "executed_tflops_per_step": ex / 1e12,
                    "achieved_tflops_whole_step": world * ex * k / e / 1e12, "frac_of_mfma_peak_whole_step": ex * k / e / 1e12 / PEAK_TFLOPS}
        side["batch8"] = side_leg(8, dt, 5)
        if world == 1:
            side["bf16" if dt == torch.float16 else "fp16"] = side_leg(1, torch.bfloat16 if dt == torch.float16 else torch.float16, args.steps)
            torch.cuda.empty_cache()
            side["torch_rocm_reference"] = torch_rocm_reference(model, x, y, dt, dev)
        for bk, key in (("batch1", None), ("batch8", "batch8")) if world == 1 else ():
            mine_ms = (1e3 * elapsed / args.steps) if key is None else side[key]["ms_per_step"]
            if bk in side["torch_rocm_reference"]:
                side["torch_rocm_reference"][bk]["this_framework_ms_per_step"] = mine_ms
                side["torch_rocm_reference"][bk]["speedup"] = side["torch_rocm_reference"][bk]["ms_per_step"] / mine_ms

    # ddim: the rest of the metric (SURVEY.md section 8d metric (2)), outside the headline's timed region: the VAE leg on
    # this GPU, and ONE whole sampling job - the 25-step DDIM loop (plan + 25 x forward_with_cfg + update) followed by the
    # decode of its sample - timed end to end like inference.py:306-348 runs it.
    dleg = None
    if decode_leg:
        R = max(1, args.repeats)
        sample = last["out"]["sample"]

        def run_decode(k):
            for _ in range(k):
                last["dec"] = pipeline.latents_to_primitives(sample, vae, mean, std)

        def run_job(k):
            for _ in range(k):
                s25 = diffusion.ddim_sample_loop(model.forward_with_cfg, tuple(x.shape), noise=x, clip_denoised=False,
                                                 model_kwargs=kw, device=dev)
                last["job"] = pipeline.latents_to_primitives(s25, vae, mean, std)

        run_decode(2)
        KD = 5
        d_all, _ = timed_repeats(run_decode, KD, R, world, dist, dev)
        j_all, _ = timed_repeats(run_job, 1, R, world, dist, dev)
        assert last["dec"].shape == (B, N, 3076) and torch.isfinite(last["dec"]).all() and torch.isfinite(last["job"]).all()
        d_s, j_s = statistics.median(d_all) / KD, statistics.median(j_all)
        dleg = {"decode": {"ms": 1e3 * d_s, "samples_per_s": world * B / d_s, "repeats_ms": [1e3 * e / KD for e in d_all],
                           "workload": f"latents_to_primitives of {B} sample(s) x {N} primitives (latent de-normalise + vae3d_dib "
                                       f"decode 1x4^3 -> 6x8^3 + inverse normalisation), {args.dtype} MFMA inputs / fp32 accumulation",
                           "algorithmic_tflops": B * N * VAE_FLOPS_PER_PRIM / 1e12,
                           "frac_of_mfma_peak": B * N * VAE_FLOPS_PER_PRIM / d_s / 1e12 / PEAK_TFLOPS},
                "samples_per_s_measured": world * B / j_s, "job_ms": 1e3 * j_s, "job_repeats_ms": [1e3 * e for e in j_all],
                "job": "one whole 25-step DDIM loop (timestep plan, 25 x forward_with_cfg + fused update) + decode of the "
                       "sample, barrier + synchronize on both sides, median of R, MAX over ranks"}
        if not args.no_kernel_events and rank == 0:
            ops.PROFILE = []
            run_decode(3)
            torch.cuda.synchronize()
            dprof, ops.PROFILE = ops.PROFILE, None
            dleg["decode"]["roofline"], dleg["decode"]["kernels"] = kernel_report(dprof, 3, traffic_file(), event_pair_overhead_ms())

    # c4: the decode leg, timed separately (median of R)
    decode_s = None
    if args.config == "c4":
        pipeline.latents_to_primitives(last["out"]["sample"], vae, mean, std)
        dl = []
        for _ in range(max(1, args.repeats)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rp = pipeline.latents_to_primitives(last["out"]["sample"], vae, mean, std)
            torch.cuda.synchronize()
            dl.append(time.perf_counter() - t0)
        assert rp.shape == (B, N, 3076) and torch.isfinite(rp).all()
        decode_s = statistics.median(dl)

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        res.update({"ms_per_step": ms_step, "repeats": len(elapsed_all),
                    "repeats_ms_per_step": [1e3 * e / args.steps for e in elapsed_all],
                    "data": "synthetic (seeded normal latents + conditioning tokens, deterministic synthetic weights "
                            "oracle/synth.py seed 4321, all layers non-zero)"})
        res["per_rank_ms_per_step"] = per_rank
        if os.environ.get("PRIMX_BENCH_SHARE_GPU") == "1":
            res["test_mode"] = "PRIMX_BENCH_SHARE_GPU=1: all ranks on one GPU over gloo - a code-path check, not a measurement"
        res["world_size_seen"] = len(ranks_seen)
        res["ranks_seen"] = ranks_seen
        res["weight_broadcast_ms"] = wb_ms                      # the collective alone (MAX over ranks), outside every timed loop
        if world > 1 and args.config in ("ddim", "c4"):
            res["weight_pack_ms_rank0"] = pack_ms
            res["weight_broadcast_GBps"] = wbytes / wb_ms / 1e6 if wb_ms else None
        par = f"batch-sharded replicas x{world}, no collective in the loop"
        if args.config == "ddim":
            steps_per_s = world * B * args.steps / elapsed
            flops_step = 2 * B * forward_flops(N, L_COND)
            res.update({
                "metric": "DiT denoise-steps/sec (N_prim=2048) + samples/sec @25-step DDIM",
                "value": steps_per_s, "unit": "denoise-steps/s",
                "config": {"workload": workload_name(B, N, args.dtype, "ddim25"),
                           "parallelism": par, "weight_broadcast_bytes": wbytes, "reuse_cond_kv": bool(args.reuse_cond_kv)},
                "samples_per_s_at_25_steps": steps_per_s / 25.0,
                "algorithmic_tflops_per_step": flops_step / 1e12})
            # Rates are quoted against the FLOPs the step EXECUTES (round-3 review): the algorithmic count of the reference's step
            # stands next to them, and the like-for-like timing of the unreduced work is `with_expanded_null_kv`.
            ex = flops_step
            if args.reuse_cond_kv:   # to_k / to_v of the conditioning tokens once per 25-step loop
                ex = flops_step - B * kv_projection_flops(L_COND) * (1.0 - 1.0 / 25.0)
            elif getattr(model, "dedup_null_kv", False) and L_COND >= 64:
                # forward_with_cfg's unconditional half: L identical conditioning rows (null_cond_embedding.expand_as(y)) - 64 + L % 64
                # of them are projected
                ln = 64 + (L_COND % 64 if L_COND > 64 else 0)
                ex = flops_step - B * kv_projection_flops(L_COND) * (1.0 - ln / L_COND)
            res.update({"executed_tflops_per_step": ex / 1e12,
                        "achieved_tflops_whole_step": world * ex * args.steps / elapsed / 1e12,
                        "frac_of_mfma_peak_whole_step": ex * args.steps / elapsed / 1e12 / PEAK_TFLOPS,
                        "flops_note": "achieved_tflops_whole_step / frac_of_mfma_peak_whole_step = EXECUTED FLOPs / time; "
                                      "algorithmic_tflops_per_step is the reference's unreduced step (SURVEY.md section 8d)"})
            if not args.reuse_cond_kv and getattr(model, "dedup_null_kv", False) and L_COND >= 64:
                res["config"]["null_cond_kv"] = (f"the unconditional half's {L_COND} identical conditioning rows are projected once "
                                                 f"({ln} rows) and addressed as the {L_COND}-key sequence by the attention kernel "
                                                 "(bit-identical results); `with_expanded_null_kv` times the expanded form")
            if (not args.reuse_cond_kv and getattr(model, "kv_ride", False) and getattr(model, "fold_ln", False) and model._fold_ok(2 * B * N, N)
                    and B == 1):
                res["config"]["cond_kv_projection"] = ("every step projects the conditioning tokens of all blocks (to_k / to_v, attention.py:106-107); block "
                                                       "i + 1's projection runs on the CUs block i's qkv launch leaves idle (primx_linear_heads_fold_pair), "
                                                       "block 0's as a launch of its own; `with_batched_kv_projection` times the one-launch form")
        elif args.config == "decode":
            res.update({
                "metric": "VAE decode samples/sec (2048 primitives per sample: latent de-normalise + vae3d_dib decode + inverse normalisation)",
                "value": world * B * args.steps / elapsed, "unit": "samples/s",
                "config": {"workload": f"second leg of BASELINE configs[3]: latents_to_primitives of {B} sample(s) x {N} "
                                       f"primitives (1x4^3 -> 6x8^3), {args.dtype} MFMA inputs / fp32 accumulation",
                           "parallelism": par, "weight_broadcast_bytes": wbytes},
                "algorithmic_tflops_per_step": B * N * VAE_FLOPS_PER_PRIM / 1e12,
                "achieved_tflops_whole_step": world * B * N * VAE_FLOPS_PER_PRIM * args.steps / elapsed / 1e12,
                "frac_of_mfma_peak_whole_step": B * N * VAE_FLOPS_PER_PRIM * args.steps / elapsed / 1e12 / PEAK_TFLOPS})
        else:
            total = 100 * elapsed / args.steps + decode_s
            res.update({
                "metric": "samples/sec, 100-step DDIM + vae3d_dib decode (BASELINE configs[3])",
                "value": world * B / total, "unit": "samples/s",
                "config": {"workload": f"BASELINE configs[3]: DiT-XL N_prim={N}, batch {B}/GPU (eff. {2 * B}), 100 DDIM steps "
                                       f"(K = {args.steps} steps timed, x100/K) + decode of {B} x {N} primitives (timed whole)",
                           "parallelism": par, "weight_broadcast_bytes": wbytes, "reuse_cond_kv": bool(args.reuse_cond_kv)},
                "ddim_ms_per_step": ms_step, "decode_ms": 1e3 * decode_s, "seconds_per_batch": total})
        if reuse:
            res["with_reuse_cond_kv"] = reuse
        if expanded:
            res["with_expanded_null_kv"] = expanded
        if unfolded:
            res["with_layernorm_launches"] = unfolded
        if batched_kv:
            res["with_batched_kv_projection"] = batched_kv
        if args.config in ("ddim", "c4"):
            res["ln_fold"] = {"enabled": bool(getattr(model, "fold_ln", False) and model._fold_ok(2 * B * N, N)),
                              "what": "LayerNorm + modulate folded into the gate-residual GEMM in front of it (operand + partial row sums) "
                                      "and the Linear behind it (statistics + per-timestep u, v in the epilogue): csrc/gemm.hip"}
        res.update(side)
        if args.config in ("ddim", "c4"):
            res["ln_in_gemm_tail"] = {"enabled": bool(getattr(model, "fuse_ln", False) and getattr(model, "ln_in_kernel", False)),
                                      "sync_timeouts": ops.ln_sync_timeouts()}
        if prof:
            res["roofline"], res["kernels"] = kernel_report(prof, args.steps, traffic_file(), ev_over)
        if dleg:
            res["decode"] = dleg["decode"]
            res["samples_per_s_measured"] = dleg["samples_per_s_measured"]
            res["measured_job"] = {k: dleg[k] for k in ("job_ms", "job_repeats_ms", "job")}

        # ------------------------------------------------------------------ parity of the benchmarked model + CPU baseline
        if world == 1 and args.config == "ddim" and not args.no_parity and N == 2048:
            parity = {}
            gfile = os.path.join(ROOT, "tests", "golden", "xl_c2.npz")
            if os.path.exists(gfile) and B == 1:
                import numpy as np
                g = np.load(gfile)
                stride = int(g["token_stride"])
                tt = torch.as_tensor(g["t"]).to(dev)
                out = model.forward_with_cfg(x, tt, y, 6.0, dt, True)
                ref = torch.from_numpy(g["forward_cfg"])
                got = out[:, ::stride].float().cpu()
                parity["rel_l2_vs_reference_fp32_28_blocks"] = float((got - ref).norm() / ref.norm())
                parity["reference"] = "tests/golden/xl_c2.npz: forward_with_cfg of the REAL reference (fp32, all 28 blocks) on these weights and inputs"
            res["parity"] = parity
        if world == 1 and not args.no_cpu_baseline:
            if args.config == "ddim":
                rec, ref, (sd14, xc, yc, tc, nb) = cpu_baseline_ddim(N)
                res["cpu_baseline"] = rec
                if not args.no_parity and N == 2048:
                    if nb == XL["depth"]:
                        mp = model                                 # (the benchmarked model itself: load_synth_weights uses the same seed)
                    else:
                        with torch.device(dev):
                            mp = pkg.DiT(**{**XL, "depth": nb}).eval()
                        mp.load_state_dict(sd14, strict=True)
                    got = mp.forward_with_cfg(xc.to(dev), tc.to(dev), yc.to(dev), 6.0, dt, True).float().cpu()
                    res.setdefault("parity", {}).update({
                        "rel_l2_vs_fp32_oracle": float((got - ref).norm() / ref.norm()), "blocks": nb,
                        "oracle": f"the cpu_baseline leg's own fp32 forward (oracle/dit_ref.py) of {nb} blocks + final layer"})
                    del mp
                if dleg:
                    rec, ref, z = cpu_baseline_decode(vae_sd, n_prims=512)
                    got = vae.decode(z.to(dev)).float().cpu()
                    res["decode"]["cpu_baseline"] = rec
                    res["decode"]["parity"] = {"max_abs_vs_fp32_oracle": float((got - ref).abs().max()),
                                               "rel_l2_vs_fp32_oracle": float((got - ref).norm() / ref.norm()),
                                               "primitives": int(z.shape[0]), "ref_abs_max": float(ref.abs().max())}
            elif args.config == "decode":
                rec, ref, z = cpu_baseline_decode(vae_sd)
                res["cpu_baseline"] = rec
                got = vae.decode(z.to(dev)).float().cpu()
                res["parity"] = {"max_abs_vs_fp32_oracle": float((got - ref).abs().max()),
                                 "rel_l2_vs_fp32_oracle": float((got - ref).norm() / ref.norm()), "primitives": int(z.shape[0]),
                                 "ref_abs_max": float(ref.abs().max())}
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
