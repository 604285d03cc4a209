"""Batch-sharded multi-GPU sampling: one process per GPU, RCCL over xGMI only at the edges.

Samples never interact on this path (attention is within one sample's tokens, CFG pairs a sample
with its own null-conditioned copy, VAE primitives are independent - SURVEY.md section 8e), so the
unit of sharding is the user sample and the DDIM loop itself needs NO collective:

    once   : broadcast(weights)   - ONE flat buffer per dtype (the reference has 515 tensors; a single
             large broadcast is link-efficient on point-to-point xGMI, 515 small ones are latency-bound)
             scatter(noise), scatter(conditioning)   (rank 0 draws the whole batch's INITIAL noise from one seeded CPU
             generator, so a seed gives the same starting latents for any world size - the role of the seeded
             `torch.randn` of inference.py:251,316.  `reference_rng=True` replays the reference CLI's DRAW ORDER instead
             (global `manual_seed`, the unused `randn(1, N, 1, 4, 4, 4)`, then the noise: `initial_noise`).  That equals the
             CLI's seed-42 latents only when nothing else consumes the global CPU generator between the seeding and the
             first image: inference.py builds model, VAE and conditioner in between (:254-256), whose constructors draw
             from it - a caller that wants the CLI's bits has to construct in the same order)
    loop   : per-rank ``ddim_sample_loop`` on its slice            (zero collectives)
    end    : gather(samples) to rank 0

Works unchanged with world_size 1 (no process group needed) and on CPU tensors with the ``gloo``
backend for the plumbing tests; the sampler itself still requires a HIP device.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous, as-even-as-possible slices: the first ``n_items % world_size`` ranks get one extra."""
    base, extra = divmod(n_items, world_size)
    out, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def _world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# Test hook: with a process group of ONE rank the collectives are skipped (nothing to exchange) - unless this is set
# (PRIMX_FORCE_COLLECTIVES=1), in which case every broadcast / scatter / gather below is really issued on the 1-rank group.
# That is how the RCCL code path is exercised on a single-GPU box (tests/test_hip_rccl.py): the library loads, the IPC mode is
# right, the packed blob survives the broadcast byte for byte.  It says nothing about scaling.
FORCE_COLLECTIVES = os.environ.get("PRIMX_FORCE_COLLECTIVES") == "1"


def _single(world: int) -> bool:
    """True when the collectives are skipped: one rank and no forcing (or no process group at all)."""
    return world == 1 and not (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def broadcast_module_(module: torch.nn.Module, src: int = 0) -> int:
    """In-place broadcast of every parameter and buffer as ONE flat tensor per dtype.  Returns bytes sent."""
    rank, world = _world()
    if _single(world):
        return 0
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    total = 0
    for dt in sorted({t.dtype for t in tensors}, key=str):
        group = [t for t in tensors if t.dtype == dt]
        flat = torch.cat([t.reshape(-1) for t in group])
        dist.broadcast(flat, src=src)
        off = 0
        for t in group:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
        total += flat.numel() * flat.element_size()
    if hasattr(module, "repack"):
        module.repack()
    return total


def broadcast_packed_(model: torch.nn.Module, dtype: torch.dtype, src: int = 0) -> int:
    """Broadcast what the 16-bit sampling path actually reads instead of the fp32 parameters: the model's packed weight
    blob (ONE flat 16-bit buffer, 1.82 GB for DiT-XL - `DiT.packed`) plus the few fp32 tensors used outside autocast
    (token / timestep embedders, null conditioning row).  Half the bytes of `broadcast_module_` over the xGMI links, no
    concatenation copy on either side (the blob IS the storage of the per-layer operands), and the receiving ranks skip
    the fp32 -> 16-bit repack.  Non-source ranks are marked packed-only: their fp32 parameters are not valid, so the
    fp32 route and a different dtype raise there.  Returns bytes sent (0 for world size 1)."""
    rank, world = _world()
    if _single(world):
        return 0
    if rank == src:
        pk = model.packed(dtype)
    else:                                                          # (an allocation the caller made ahead of the collective is reused)
        pk = next((v for (d, _), v in getattr(model, "_pack", {}).items() if d == dtype), None) or model.packed_alloc(dtype)
    dist.broadcast(pk["_flat"], src=src)
    small = model.small_fp32_tensors()
    flat32 = torch.cat([t.reshape(-1).float() for t in small])
    dist.broadcast(flat32, src=src)
    if rank != src:
        off = 0
        for t in small:
            n = t.numel()
            t.copy_(flat32[off:off + n].view_as(t))
            off += n
        model._packed_only = True
    return pk["_flat"].numel() * pk["_flat"].element_size() + flat32.numel() * 4


def scatter_batch(full: Optional[torch.Tensor], shape_tail: Sequence[int], n_items: int, dtype: torch.dtype,
                  device, src: int = 0) -> torch.Tensor:
    """Rank ``src`` holds ``full`` [n_items, *shape_tail]; every rank receives its contiguous slice."""
    rank, world = _world()
    lo, hi = shard_bounds(n_items, world)[rank]
    if _single(world):
        return full[lo:hi].to(device=device, dtype=dtype)
    mine = torch.empty((hi - lo, *shape_tail), dtype=dtype, device=device)
    # uneven slices: pad every chunk to the largest so a single scatter suffices
    mx = max(h - l for l, h in shard_bounds(n_items, world))
    recv = torch.empty((mx, *shape_tail), dtype=dtype, device=device)
    chunks = None
    if rank == src:
        chunks = []
        for l, h in shard_bounds(n_items, world):
            c = torch.zeros((mx, *shape_tail), dtype=dtype, device=device)
            c[:h - l] = full[l:h].to(device=device, dtype=dtype)
            chunks.append(c)
    dist.scatter(recv, chunks, src=src)
    mine.copy_(recv[:hi - lo])
    return mine


def gather_batch(local: torch.Tensor, n_items: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Inverse of scatter_batch: rank ``dst`` gets [n_items, ...], other ranks None."""
    rank, world = _world()
    if _single(world):
        return local
    bounds = shard_bounds(n_items, world)
    mx = max(h - l for l, h in bounds)
    send = torch.zeros((mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
    send[:local.shape[0]] = local
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:h - l] for b, (l, h) in zip(bufs, bounds)], dim=0)


def initial_noise(batch: int, n_tokens: int, channels: int, seed: Optional[int], reference_rng: bool = False) -> torch.Tensor:
    """The [batch, n_tokens, channels] starting latents, drawn on the CPU by rank 0.

    Default: a private generator seeded with `seed` (no global state touched) - reproducible for any world size.

    `reference_rng=True`: the draws of the reference CLI, in its order and from the process-GLOBAL CPU generator
    (inference.py:251,313,316): `torch.manual_seed(seed)` (skipped when `seed` is None: the caller seeded, as inference.py does
    once per process, and every call continues the stream like the CLI's per-image loop), then the unused
    `torch.randn(1, n_tokens, 1, 4, 4, 4)`, then `torch.randn(batch, n_tokens, channels)`.  With batch = 1 (`inf_bs` of the CLI) the
    latents are the CLI's for the same seed and the same draws before them (the CLI constructs its modules between the seeding and
    the first image: a caller that wants its numbers seeds, builds the models in that order, and passes seed=None here); for a
    larger batch entry 0 is still the CLI's sample whenever n_tokens * channels is a multiple of 16 (torch fills normal draws
    in blocks of 16)."""
    if not reference_rng:
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        return torch.randn(batch, n_tokens, channels, generator=gen)              # CPU draw, as inference.py:316
    if seed is not None:
        torch.manual_seed(seed)
    torch.randn(1, n_tokens, 1, 4, 4, 4)                                           # inference.py:313 (`latent`: only its shape is used)
    return torch.randn(batch, n_tokens, channels)


class ShardedSampler:
    """Batch-sharded DDIM sampling over the ranks of the default process group."""

    def __init__(self, model: torch.nn.Module, diffusion, device, sync_weights: bool = True,
                 packed_dtype: Optional[torch.dtype] = None):
        """sync_weights: broadcast rank 0's weights at construction - the packed 16-bit blob when `packed_dtype` is given
        (the sampling dtype; half the bytes, receivers need no repack), else every fp32 parameter."""
        self.model, self.diffusion, self.device = model, diffusion, torch.device(device)
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)   # RCCL and the primx_* launches use the current device's streams
        self.rank, self.world = _world()
        self.weight_bytes = 0
        if sync_weights:
            self.weight_bytes = (broadcast_packed_(model, packed_dtype, 0) if packed_dtype is not None
                                 else broadcast_module_(model, 0))

    def sample(self, batch: int, n_tokens: int, channels: int, cond: Optional[torch.Tensor], seed: Optional[int],
               loop: Optional[Callable] = None, reference_rng: bool = False, **model_kwargs) -> Optional[torch.Tensor]:
        """cond: [batch, L, Dc] on rank 0 (None elsewhere).  Returns the [batch, n_tokens, channels] samples on
        rank 0.  ``loop(noise_slice, cond_slice) -> samples`` overrides the DDIM call (used by the CPU tests).
        ``reference_rng``: draw the starting latents exactly as the reference CLI does (`initial_noise`)."""
        noise = None
        if self.rank == 0:
            noise = initial_noise(batch, n_tokens, channels, seed, reference_rng)
        cond_tail = None
        if not _single(self.world):
            meta = [tuple(cond.shape[1:])] if self.rank == 0 else [None]
            dist.broadcast_object_list(meta, src=0)
            cond_tail = meta[0]
        else:
            cond_tail = tuple(cond.shape[1:])
        x = scatter_batch(noise, (n_tokens, channels), batch, torch.float32, self.device)
        y = scatter_batch(cond, cond_tail, batch, torch.float32, self.device)
        if x.shape[0] == 0:
            out = x
        elif loop is not None:
            out = loop(x, y)
        else:
            out = self.diffusion.ddim_sample_loop(self.model.forward_with_cfg, tuple(x.shape), noise=x,
                                                  clip_denoised=False, model_kwargs=dict(y=y, **model_kwargs),
                                                  device=self.device)
        return gather_batch(out, batch)

    def sample_and_decode(self, batch: int, n_tokens: int, channels: int, cond: Optional[torch.Tensor],
                          seed: Optional[int], decode: Callable[[torch.Tensor], torch.Tensor],
                          loop: Optional[Callable] = None, reference_rng: bool = False, **model_kwargs) -> Optional[torch.Tensor]:
        """`sample`, then every rank decodes ITS OWN samples (`decode(samples [b, N, C]) -> [b, N, F]`, e.g.
        `lambda s: pipeline.latents_to_primitives(s, vae, mean, std)`: primitives are independent, SURVEY.md section
        8e) and only the decoded primitives are gathered on rank 0 - the 25 MB/sample payload crosses xGMI once, the
        decoder work is spread over all GPUs."""
        noise = None
        if self.rank == 0:
            noise = initial_noise(batch, n_tokens, channels, seed, reference_rng)
        if not _single(self.world):
            meta = [tuple(cond.shape[1:])] if self.rank == 0 else [None]
            dist.broadcast_object_list(meta, src=0)
            cond_tail = meta[0]
        else:
            cond_tail = tuple(cond.shape[1:])
        x = scatter_batch(noise, (n_tokens, channels), batch, torch.float32, self.device)
        y = scatter_batch(cond, cond_tail, batch, torch.float32, self.device)
        if x.shape[0] == 0:
            dec = None
        else:
            if loop is not None:
                out = loop(x, y)
            else:
                out = self.diffusion.ddim_sample_loop(self.model.forward_with_cfg, tuple(x.shape), noise=x,
                                                      clip_denoised=False, model_kwargs=dict(y=y, **model_kwargs),
                                                      device=self.device)
            dec = decode(out)
        if _single(self.world):
            if dec is None:                                    # batch == 0: an empty result, not None
                return torch.empty(0, n_tokens, 0, dtype=torch.float32, device=self.device)
            return dec
        # ranks without samples (batch < world size) still take part in the gather: they learn the WHOLE per-sample shape
        # of the decoder's output first (a `decode` may reshape, e.g. [b, N, C, S, S, S]); rank 0 holds samples whenever
        # batch > 0 (shard_sizes gives the first ranks the larger shares)
        tail = [tuple(int(d) for d in dec.shape[1:]) if dec is not None else None] if self.rank == 0 else [None]
        dist.broadcast_object_list(tail, src=0)
        if dec is None:
            dec = torch.empty(0, *(tail[0] if tail[0] is not None else (n_tokens, 0)), dtype=torch.float32, device=self.device)
        return gather_batch(dec.float().contiguous(), batch)
