"""N > 1 path on CPU: two gloo ranks exercise the sharding plumbing (flat weight broadcast, uneven
scatter of noise/conditioning, collective-free per-rank loop, gather) around a stand-in loop."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, batch, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import topia_xl_amd as pkg
        from topia_xl_amd.sharding import ShardedSampler, shard_bounds
        torch.manual_seed(100 + rank)                                   # ranks start with DIFFERENT weights
        model = pkg.DiT(seq_length=8, in_channels=4, condition_channels=8, hidden_size=64, depth=1, num_heads=2,
                        cond_drop_prob=0.1, attn_proj_bias=True)
        d = pkg.create_diffusion("ddim5", noise_schedule="squaredcos_cap_v2", parameterization="v")
        sampler = ShardedSampler(model, d, "cpu")
        assert (sampler.weight_bytes > 0) == (world > 1)
        digest = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double().sum()
        all_d = [torch.zeros((), dtype=torch.double) for _ in range(world)]
        dist.all_gather(all_d, digest)
        assert all(torch.equal(all_d[0], x) for x in all_d), "weights differ after broadcast"
        cond = torch.arange(batch * 3 * 8, dtype=torch.float32).reshape(batch, 3, 8) if rank == 0 else None
        seen = {}

        def loop(x, y):                                                 # stand-in for the DDIM loop: no collectives
            seen["n"] = x.shape[0]
            return x * 2 + y[:, :1, :4].sum(-1, keepdim=True)

        out = sampler.sample(batch, 8, 4, cond, seed=42, loop=loop)
        lo, hi = shard_bounds(batch, world)[rank]
        assert seen.get("n", 0) == hi - lo
        if rank == 0:
            noise = torch.randn(batch, 8, 4, generator=torch.Generator().manual_seed(42))
            want = noise * 2 + cond[:, :1, :4].sum(-1, keepdim=True)
            assert out.shape == (batch, 8, 4) and torch.equal(out, want)
        else:
            assert out is None
        # reference_rng: rank 0 replays the reference CLI's draws (inference.py:251,313,316) - global seed, the unused
        # randn(1, N, 1, 4, 4, 4), then the noise - and the shards reassemble to exactly those latents
        out = sampler.sample(batch, 8, 4, cond, seed=42, loop=lambda x, y: x, reference_rng=True)
        if rank == 0:
            torch.manual_seed(42)
            torch.randn(1, 8, 1, 4, 4, 4)
            assert torch.equal(out, torch.randn(batch, 8, 4))
            ret.put("ok")
    finally:
        dist.destroy_process_group()


def _worker_packed(rank, world, port, batch, ret):
    """Packed-blob broadcast (what the 16-bit path reads: ONE flat 16-bit buffer + the small fp32 tensors) and the
    sharded decode: every rank decodes its own samples, only decoded primitives are gathered."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import topia_xl_amd as pkg
        from topia_xl_amd.sharding import ShardedSampler
        torch.manual_seed(200 + rank)
        cfg = dict(seq_length=8, in_channels=4, condition_channels=8, hidden_size=64, depth=2, num_heads=2,
                   cond_drop_prob=0.1, attn_proj_bias=True)
        model = pkg.DiT(**cfg)
        for p in model.parameters():                                    # no zero-initialised layers: a real digest
            torch.nn.init.normal_(p, std=0.1)
        model.repack()
        ref_flat = model.packed(torch.float16)["_flat"].clone() if rank == 0 else None
        model.repack()
        d = pkg.create_diffusion("ddim5", noise_schedule="squaredcos_cap_v2", parameterization="v")
        sampler = ShardedSampler(model, d, "cpu", packed_dtype=torch.float16)
        pk = model.packed(torch.float16)                                # cached: the received blob on rank 1
        n16 = pk["_flat"].numel()
        small = sum(t.numel() for t in model.small_fp32_tensors())
        assert sampler.weight_bytes == 2 * n16 + 4 * small
        assert sampler.weight_bytes < 0.6 * 4 * sum(p.numel() for p in model.parameters())   # ~half of the fp32 broadcast
        assert model._packed_only == (rank != 0)
        # every operand view aliases the blob (no second copy), in blob order
        assert pk["blocks"][1]["w_fc2"].data_ptr() >= pk["_flat"].data_ptr() and pk["w_final"].shape == (8, 64)
        dig = torch.stack([pk["_flat"].double().sum(), torch.cat([t.reshape(-1) for t in model.small_fp32_tensors()]).double().sum()])
        all_d = [torch.zeros(2, dtype=torch.double) for _ in range(world)]
        dist.all_gather(all_d, dig)
        assert all(torch.equal(all_d[0], x) for x in all_d), "packed weights differ after broadcast"
        if rank == 0:
            assert torch.equal(pk["_flat"], ref_flat)
        else:
            with pytest.raises(RuntimeError, match="packed"):
                model.packed(torch.bfloat16)
        cond = torch.arange(batch * 3 * 8, dtype=torch.float32).reshape(batch, 3, 8) if rank == 0 else None
        calls = {}

        def decode(samples):                                            # stand-in for latents_to_primitives
            calls["n"] = samples.shape[0]
            # a decoder may reshape (e.g. [b, N, C, S, S, S]): the token dim is NOT kept - the ranks without samples
            # (batch 1 on two ranks) must learn the whole per-sample shape, not just the last dim
            return torch.cat([samples, samples.sum(-1, keepdim=True)], dim=-1).view(samples.shape[0], 2, 4, 5)

        out = sampler.sample_and_decode(batch, 8, 4, cond, seed=7, decode=decode, loop=lambda x, y: x + y[:, :1, :4])
        from topia_xl_amd.sharding import shard_bounds
        lo, hi = shard_bounds(batch, world)[rank]
        assert calls.get("n", 0) == hi - lo
        if rank == 0:
            noise = torch.randn(batch, 8, 4, generator=torch.Generator().manual_seed(7))
            s = noise + cond[:, :1, :4]
            assert out.shape == (batch, 2, 4, 5) and torch.equal(out, torch.cat([s, s.sum(-1, keepdim=True)], dim=-1).view(batch, 2, 4, 5))
            ret.put("ok")
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [3, 1])
def test_two_rank_packed_broadcast_and_sharded_decode(batch):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + batch
    procs = [ctx.Process(target=_worker_packed, args=(r, 2, port, batch, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == "ok"


@pytest.mark.parametrize("batch", [4, 5, 1])
def test_two_rank_sharded_sampling(batch):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + batch
    procs = [ctx.Process(target=_worker, args=(r, 2, port, batch, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == "ok"


def test_reference_rng_replays_the_cli_draw_order():
    """`initial_noise(reference_rng=True)` = the reference CLI's draws in its ORDER (inference.py:251: torch.manual_seed(seed);
    :313 latent = torch.randn(1, num_prims, 1, 4, 4, 4); :316 inf_x = torch.randn(inf_bs, num_prims, 68)): given the same state of
    the global CPU generator it produces the same latents (the CLI's own bits additionally depend on what its model / VAE /
    conditioner constructors draw between the seeding and the first image, inference.py:254-256 - not replayed here); a second
    call continues the stream like the CLI's per-image loop; a larger batch keeps entry 0; the default (private generator) leaves
    the global stream alone and gives other numbers."""
    sys.path.insert(0, ROOT)
    from topia_xl_amd.sharding import initial_noise
    N = 2048
    torch.manual_seed(42)
    torch.randn(1, N, 1, 4, 4, 4)
    cli_first = torch.randn(1, N, 68)
    torch.randn(1, N, 1, 4, 4, 4)
    cli_second = torch.randn(1, N, 68)
    assert torch.equal(initial_noise(1, N, 68, 42, reference_rng=True), cli_first)
    assert torch.equal(initial_noise(1, N, 68, None, reference_rng=True), cli_second)      # the next image of the same process
    b4 = initial_noise(4, N, 68, 42, reference_rng=True)
    assert torch.equal(b4[:1], cli_first)
    torch.manual_seed(7)
    probe = torch.rand(1)
    torch.manual_seed(7)
    private = initial_noise(1, N, 68, 42)
    assert torch.equal(torch.rand(1), probe) and not torch.equal(private, cli_first)


def test_shard_bounds():
    from topia_xl_amd.sharding import shard_bounds
    assert shard_bounds(64, 8) == [(8 * i, 8 * i + 8) for i in range(8)]
    assert shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert shard_bounds(1, 4) == [(0, 1), (1, 1), (1, 1), (1, 1)]
    b = shard_bounds(16, 4)
    assert b[0][0] == 0 and b[-1][1] == 16 and all(b[i][1] == b[i + 1][0] for i in range(3))
