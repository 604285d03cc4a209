"""Host bookkeeping of the product (3dtopia-xl_amd/diffusion): bit-exact against the reference's
tables (golden) and against the oracle's float32 per-step arithmetic."""
import numpy as np
import pytest
import torch

from oracle import diffusion_ref as dref


def test_tables_and_maps_bit_exact(pkg, golden):
    g = golden("schedule")
    for n in (5, 25, 50, 100, 200):
        d = pkg.create_diffusion(timestep_respacing=f"ddim{n}", noise_schedule="squaredcos_cap_v2",
                                 parameterization="v", diffusion_steps=1000)
        assert d.num_timesteps == n
        assert d.timestep_map == g[f"ddim{n}_map"].tolist()           # integer, bit-exact
        assert sorted(d.use_timesteps) == d.timestep_map
        for attr in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                     "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                     "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
                     "posterior_mean_coef2"):
            assert np.array_equal(getattr(d, attr), g[f"ddim{n}_{attr}"]), (n, attr)
    from topia_xl_amd.diffusion import get_named_beta_schedule
    assert np.array_equal(get_named_beta_schedule("squaredcos_cap_v2", 1000), g["cos1000_betas"])
    assert np.array_equal(get_named_beta_schedule("linear", 1000), g["lin1000_betas"])
    assert np.array_equal(get_named_beta_schedule("linear", 250), g["lin250_betas"])
    with pytest.raises(NotImplementedError):
        get_named_beta_schedule("nope", 10)


def test_space_timesteps(pkg, golden):
    g = golden("schedule")
    assert sorted(pkg.space_timesteps(300, [10, 15, 20])) == g["sections_300_10_15_20"].tolist()
    assert sorted(pkg.space_timesteps(1000, "7,3,11")) == g["sections_1000_str"].tolist()
    assert sorted(pkg.space_timesteps(1000, "ddim25")) == list(range(0, 1000, 40))
    with pytest.raises(ValueError):
        pkg.space_timesteps(1000, "ddim600")        # no integer stride gives exactly 600 steps
    with pytest.raises(ValueError):
        pkg.space_timesteps(10, [20])
    full = pkg.create_diffusion(timestep_respacing="", noise_schedule="linear", parameterization="eps",
                                learn_sigma=False, diffusion_steps=50)
    assert full.timestep_map == g["full50_map"].tolist()
    assert np.array_equal(full.alphas_cumprod, g["full50_alphas_cumprod"])
    with pytest.raises(NotImplementedError):
        pkg.create_diffusion("ddim5", parameterization="bogus")


@pytest.mark.parametrize("n,eta", [(5, 0.0), (25, 0.0), (25, 0.7), (100, 1.0)])
def test_step_coefficients_match_float32_reference_ops(pkg, n, eta):
    """Every scalar of the device table equals what the reference's float32 tensor ops produce
    (evaluated here with torch CPU ops through the oracle's tables) - bit for bit."""
    d = pkg.create_diffusion(f"ddim{n}", noise_schedule="squaredcos_cap_v2", parameterization="v")
    tab, _ = dref.make("squaredcos_cap_v2", 1000, f"ddim{n}")
    c = d.step_coefficients(eta)
    assert c.shape == (n, 16) and c.dtype == np.float32
    one = torch.ones(1)
    for i in range(n):
        ex = lambda arr: torch.full((1,), float(np.float32(arr[i])))
        abar, abar_prev = ex(tab.acp), ex(tab.acp_prev)
        sq = dref._sqrt32   # IEEE-correct fp32 sqrt (what the reference's CUDA device computes)
        sigma = eta * sq((1 - abar_prev) / (1 - abar)) * sq(1 - abar / abar_prev)
        want = {0: ex(tab.sqrt_acp), 1: ex(tab.sqrt_1m_acp), 2: ex(tab.sqrt_recip_acp), 3: ex(tab.sqrt_recipm1_acp),
                4: ex(tab.post_c1), 5: ex(tab.post_c2), 6: ex(tab.post_logvar_clipped), 7: ex(np.log(tab.betas)),
                9: sq(abar_prev), 10: sq(1 - abar_prev - sigma ** 2), 11: sigma,
                12: one * (0.0 if i == 0 else 1.0)}
        for col, w in want.items():
            assert np.float32(w.item()) == c[i, col] or (np.isnan(w.item()) and np.isnan(c[i, col])), (i, col)


def test_sampler_refuses_cpu_tensors(pkg):
    d = pkg.create_diffusion("ddim5", noise_schedule="squaredcos_cap_v2", parameterization="v")
    with pytest.raises(RuntimeError, match="HIP device"):
        list(d.ddim_sample_loop_progressive(lambda x, t: x, (1, 4, 2), noise=torch.zeros(1, 4, 2), device="cpu"))
    with pytest.raises(NotImplementedError):
        list(d.ddim_sample_loop_progressive(lambda x, t: x, (1, 4, 2), noise=torch.zeros(1, 4, 2), cond_fn=lambda: 0))
