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
            ret.put("ok")
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


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


def test_shard_bounds():
    from topia_xl_amd.sharding import shard_bounds
    assert shard_bounds(64, 8) == [(8 * i, 8 * i + 8) for i in range(8)]
    assert shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert shard_bounds(1, 4) == [(0, 1), (1, 1), (1, 1), (1, 1)]
    b = shard_bounds(16, 4)
    assert b[0][0] == 0 and b[-1][1] == 16 and all(b[i][1] == b[i + 1][0] for i in range(3))
