"""The whole data flow of inference.py:312-352 at smoke size (examples/generate.py --small): every stage runs on the HIP
path and hands the next one tensors of the reference's shapes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generate_small():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "generate.py"), "--small", "--steps", "5", "--res", "48",
                        "--lattice", "24"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert "tokens (1, 17, 96)" in last and "samples (1, 64, 68)" in last and "recon_param (1, 64, 3076)" in last
    assert "sdf grid (24, 24, 24)" in last and "preview (1, 4, 48, 48)" in last


@pytest.mark.parametrize("flag", ["--selftest", "--selftest-xl"])
def test_validate_checkpoint_script(flag):
    """tools/validate_checkpoint.py (the real-checkpoint comparison to run where the released files exist) on a SYNTHETIC small
    fp16 checkpoint: checkpoint -> strict load / packed route -> forward_with_cfg, a 3-step trajectory and the VAE decode against
    the fp32 CPU side, one JSON report that says PASS."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "validate_checkpoint.py"), flag], capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    rep = json.loads(r.stdout[r.stdout.index("{"):])
    assert rep["pass"] and rep["forward_with_cfg"]["packed_route_bit_identical"] and rep["vae_decode"]["pass"]
    assert len(rep["ddim_trajectory"]["rel_l2_per_step"]) == (3 if flag == "--selftest" else 2)
    # the dynamic-range report (round 5): per block and dtype, finite operands, folded == unfolded up to rounding
    for name in ("fp16", "bf16"):
        dr = rep["dynamic_range"][name]
        assert dr["operands_finite"] and len(dr["per_block"]) == rep["config"]["depth"] and dr["residual_abs_max"] > 0
        assert dr["row_std_range"][0] > 0 and dr["operand_abs_max_unfolded"] > 0
        if flag == "--selftest-xl" and not any(os.environ.get(v) for v in ("PRIMX_DIT_FOLD", "PRIMX_GEMM_NOBIG", "PRIMX_GEMM_LOADER",
                                                                           "PRIMX_GEMM_BIGHEADS_MIN", "PRIMX_DIT_FUSE_LN", "PRIMX_CFG_STREAMS")):
            # the released model's width: planned forwards fold, and the fold's operand is as normalised as the LayerNorm output
            assert dr["fold_active"] and 0 < dr["folded_vs_unfolded_rel_l2"] < (4e-3 if name == "fp16" else 3e-2)
            assert dr["operand_abs_max_folded"] < 2 * dr["operand_abs_max_unfolded"]
