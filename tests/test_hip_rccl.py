"""RCCL on the real device (round 4): the multi-GPU path of SURVEY.md section 8e, exercised on a ONE-rank `nccl` process group.

The GPU box of the test run has one MI355X, so nothing here measures scaling.  What it proves: librccl loads and initialises
with the IPC mode this repository exports (`HSA_ENABLE_IPC_MODE_LEGACY=0`), the packed-weight broadcast (ONE flat 16-bit
buffer, 1.82 GB at DiT-XL) is byte-exact on device memory, scatter / gather of latents run on device tensors, and
`ShardedSampler.sample_and_decode` gives the same samples through the collective code path as through the plain loop
(`sharding.FORCE_COLLECTIVES`: with one rank the collectives are normally skipped).  Each case runs in its own process, so
the process group never leaks into other tests; `python bench.py --gpus 1` under `torch.distributed.run` is the last one.
"""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _env():
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               PRIMX_FORCE_COLLECTIVES="1", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _run(code: str, timeout: int = 600) -> str:
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=_env(), cwd=ROOT, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


def test_packed_weight_broadcast_and_batch_collectives_on_rccl():
    out = _run("""
        import torch, torch.distributed as dist
        import __graft_entry__; __graft_entry__.build()
        import topia_xl_amd as pkg
        from topia_xl_amd import sharding
        from oracle import synth
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        assert sharding.FORCE_COLLECTIVES and dist.get_backend() == "nccl"
        cfg = dict(in_channels=68, condition_channels=768, hidden_size=1152, depth=28)
        with torch.device(dev):
            m = pkg.DiT(seq_length=2048, num_heads=16, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
        m.load_state_dict(synth.dit_state_dict(4321, **cfg), strict=True)
        flat = m.packed(torch.float16)["_flat"]
        before = flat.view(torch.int16).clone()
        small = [t.clone() for t in m.small_fp32_tensors()]
        sent = sharding.broadcast_packed_(m, torch.float16, 0)
        torch.cuda.synchronize()
        assert sent == flat.numel() * 2 + sum(t.numel() for t in small) * 4 and sent > 1.8e9, sent
        assert torch.equal(before, m.packed(torch.float16)["_flat"].view(torch.int16))          # byte-exact on the device
        assert all(torch.equal(a, b) for a, b in zip(small, m.small_fp32_tensors()))
        full = torch.randn(5, 2048, 68)
        mine = sharding.scatter_batch(full, (2048, 68), 5, torch.float32, dev)
        assert mine.is_cuda and torch.equal(mine.cpu(), full)
        back = sharding.gather_batch(mine * 2, 5)
        assert back.is_cuda and torch.equal(back.cpu(), full * 2)
        print("RCCL_OK", sent)
        dist.destroy_process_group()
    """)
    assert "RCCL_OK" in out, out


def test_sharded_sampler_through_the_collective_path_matches_the_plain_loop():
    out = _run("""
        import torch, torch.distributed as dist
        import __graft_entry__; __graft_entry__.build()
        import topia_xl_amd as pkg
        from topia_xl_amd import sharding, pipeline
        from oracle import synth
        from tests.golden.make_golden import DIT_CASES, SEED, VAE_CFG
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        name, cfg, heads, N, L, B = DIT_CASES[1]
        m = pkg.DiT(seq_length=N, num_heads=heads, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
        m.load_state_dict(synth.dit_state_dict(SEED, **cfg), strict=True)
        m.to(dev)
        vae = pkg.VAE(**VAE_CFG).eval()
        vae.load_state_dict(synth.state_dict_like(SEED, vae.state_dict()), strict=True)
        vae.to(dev)
        d = pkg.create_diffusion("ddim5", noise_schedule="squaredcos_cap_v2", parameterization="v")
        cond = synth.tensor(SEED, "rccl.y", (3, L, cfg["condition_channels"]))
        mean, std = [0.0] * 68, [1.0] * 68
        decode = lambda s: pipeline.latents_to_primitives(s, vae, mean, std)
        kw = dict(cfg_scale=6.0, precision_dtype=torch.float16, enable_amp=True)
        sharding.FORCE_COLLECTIVES = False
        plain = sharding.ShardedSampler(m, d, dev, sync_weights=False).sample_and_decode(3, N, 68, cond, 7, decode, **kw)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        sharding.FORCE_COLLECTIVES = True
        s = sharding.ShardedSampler(m, d, dev, sync_weights=True, packed_dtype=torch.float16)
        assert s.weight_bytes > 0
        coll = s.sample_and_decode(3, N, 68, cond, 7, decode, **kw)
        assert coll.shape == plain.shape == (3, N, 4 + 6 * 512) and torch.equal(coll, plain)
        smp = s.sample(3, N, 68, cond, 7, **kw)
        assert smp.shape == (3, N, 68) and torch.isfinite(smp).all()
        print("SHARDED_OK", s.weight_bytes)
        dist.destroy_process_group()
    """)
    assert "SHARDED_OK" in out, out


def test_bench_under_the_launcher_with_one_rank_on_rccl():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` - the command form the driver uses for N > 1 - with
    the weight broadcast forced onto the 1-rank RCCL group: one JSON line, the headline fields, and the broadcast really sent
    the packed blob."""
    env = _env()
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--repeats", "1",
           "--no-cpu-baseline", "--no-parity", "--no-decode-leg", "--no-side-legs", "--no-kernel-events"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["unit"] == "denoise-steps/s" and rec["value"] > 0
    assert rec["config"]["weight_broadcast_bytes"] > 1.8e9, rec["config"]


def test_bench_two_ranks_share_one_gpu_over_gloo():
    """The N > 1 code path of bench.py on a ONE-GPU box (PRIMX_BENCH_SHARE_GPU=1: both ranks on cuda:0, gloo process group - RCCL refuses
    two ranks on one device): rank 1 receives the packed blob into a packed-only model, both ranks run their own loop, the line
    carries what the driver's SCALE record is audited with - two ranks seen by an all_gather, two per-rank times, the broadcast's bytes
    and wall time - and `value` is the whole job's rate (2 x batch x steps / MAX time).  Measures nothing."""
    env = dict(os.environ)
    env.update(PRIMX_BENCH_SHARE_GPU="1", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "PRIMX_FORCE_COLLECTIVES"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "1",
           "--no-cpu-baseline", "--no-parity", "--no-decode-leg", "--no-kernel-events", "--no-side-legs"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_seen"] == 2 and sorted(x[0] for x in d["ranks_seen"]) == [0, 1]
    assert len(d["per_rank_ms_per_step"]["all"]) == 2 and d["per_rank_ms_per_step"]["max"] >= d["per_rank_ms_per_step"]["min"] > 0
    assert d["config"]["weight_broadcast_bytes"] > 1.8e9 and d["weight_broadcast_ms"] > 0 and "test_mode" in d
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"] and d["scaling"] == "weak"
    assert d["ms_per_step"] >= d["per_rank_ms_per_step"]["max"] * 0.999
