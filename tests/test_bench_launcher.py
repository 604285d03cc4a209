"""bench.py's own launcher (CPU, gloo): `python bench.py --gpus N` with no WORLD_SIZE in the environment must start N ranks
itself (torch.distributed.run on 127.0.0.1), every rank must meet the others, and the line must say n_gpus = N - a silent
one-rank run would make a scaling curve measure nothing.  `--dry-run` stops after the rendezvous, so no GPU is needed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=env,
                          timeout=timeout, cwd=ROOT)


def _json_line(out: str):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out          # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_by_itself():
    r = _run("--gpus", "2", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _json_line(r.stdout)
    assert rec["dry_run"] is True and rec["n_gpus"] == 2
    assert sorted(x[0] for x in rec["ranks"]) == [0, 1] and sorted(x[1] for x in rec["ranks"]) == [0, 1]


def test_single_rank_needs_no_launcher():
    r = _run("--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_rank_count_mismatch_fails_loudly():
    # a launcher that started ONE rank while --gpus says 2: refuse instead of reporting a 1-GPU number as a 2-GPU one
    r = _run("--gpus", "2", "--dry-run", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_workload_names_follow_the_arguments():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.workload_name(1, 2048, "fp16", "ddim25").startswith("BASELINE configs[1]")
    assert bench.workload_name(8, 2048, "fp16", "ddim25").startswith("BASELINE configs[2] per-GPU shape")
    assert bench.workload_name(4, 4096, "bf16", "ddim25").startswith("BASELINE configs[4] per-GPU shape")
    assert bench.workload_name(1, 2048, "bf16", "ddim25").startswith("not a BASELINE configuration")
    assert bench.workload_name(2, 1024, "fp16", "ddim25").startswith("not a BASELINE configuration")


def test_kernel_report_corrects_the_event_readings(tmp_path):
    """`kernel_report`: every per-launch reading has HALF the empty event-pair reading subtracted (never more than half the reading
    itself), the dominant kernel is chosen among the MFMA kernels by corrected time, and the sums stand in the record - so that
    kernels_sum_ms_per_step <= ms_per_step can be checked from the line."""
    sys.path.insert(0, ROOT)
    import bench

    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    prof = []
    for _ in range(4):                                     # two steps x (2 GEMM launches + 1 row kernel)
        prof.append(("gemmA 8x8x8", 1e9, Ev(0.0), Ev(0.050)))
    for _ in range(2):
        prof.append(("rowop", 0.0, Ev(0.0), Ev(0.004)))
        prof.append(("gemmB 4x4x4", 2e9, Ev(0.0), Ev(0.030)))
    roof, kernels = bench.kernel_report(prof, 2, str(tmp_path / "none.json"), overhead_ms=0.006)
    assert roof["kernel"] == "gemmA 8x8x8" and roof["launches"] == 4
    assert abs(roof["avg_launch_ms"] - 0.047) < 1e-12 and abs(roof["avg_launch_ms_raw_event_reading"] - 0.050) < 1e-12
    assert abs(kernels["rowop"]["ms_per_step"] - 0.002) < 1e-12          # 4 us reading, 3 us half-pair: capped at half the reading
    assert abs(kernels["gemmB 4x4x4"]["ms_per_step"] - 0.027) < 1e-12
    assert abs(roof["kernels_sum_ms_per_step"] - (4 * 0.047 + 2 * 0.002 + 2 * 0.027) / 2) < 1e-12
    assert abs(roof["kernels_sum_ms_per_step_raw"] - (4 * 0.050 + 2 * 0.004 + 2 * 0.030) / 2) < 1e-12
    assert roof["traffic"] is None and abs(roof["achieved"] - 1e9 / 0.047e-3 / 1e12) < 1e-9
