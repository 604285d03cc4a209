"""CPU restatement of the sampler bookkeeping and update formulas (TEST INFRASTRUCTURE).

Schedule tables in numpy float64 (gaussian_diffusion.py:99-142,154-202; respace.py:12-87) and the
per-step update in torch-CPU fp32 tensor ops in the reference's operation order
(gaussian_diffusion.py:255-356, 394-435, 531-578, 880-892).  Written independently of
``3dtopia-xl_amd/diffusion`` (vectorised tables, explicit per-step functions) so that agreement
between the two is evidence, not tautology.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List

import numpy as np
import torch


def cosine_betas(T: int, max_beta: float = 0.999) -> np.ndarray:
    """squaredcos_cap_v2 (gaussian_diffusion.py:118-142).  ``math.cos`` per element like the reference
    (np.cos may differ in the last ulp on some libm builds)."""
    def abar(u: float) -> float:
        return math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
    return np.array([min(1 - abar((i + 1) / T) / abar(i / T), max_beta) for i in range(T)], dtype=np.float64)


def linear_betas(T: int) -> np.ndarray:
    """gaussian_diffusion.py:107-116."""
    s = 1000 / T
    return np.linspace(s * 0.0001, s * 0.02, T, dtype=np.float64)


def ddim_steps(T: int, n: int) -> List[int]:
    """'ddimN' striding (respace.py:32-39): first integer stride giving exactly n steps."""
    for stride in range(1, T):
        kept = list(range(0, T, stride))
        if len(kept) == n:
            return kept
    raise ValueError(f"cannot create exactly {T} steps with an integer stride")


class Tables:
    """float64 tables of a (possibly respaced) process."""

    def __init__(self, betas: np.ndarray):
        b = np.asarray(betas, dtype=np.float64)
        a = 1.0 - b
        self.betas = b
        self.n = len(b)
        self.acp = np.cumprod(a)
        self.acp_prev = np.concatenate([[1.0], self.acp[:-1]])
        self.sqrt_acp = np.sqrt(self.acp)
        self.sqrt_1m_acp = np.sqrt(1.0 - self.acp)
        self.sqrt_recip_acp = np.sqrt(1.0 / self.acp)
        self.sqrt_recipm1_acp = np.sqrt(1.0 / self.acp - 1)
        self.post_var = b * (1.0 - self.acp_prev) / (1.0 - self.acp)
        self.post_logvar_clipped = np.log(np.concatenate([self.post_var[1:2], self.post_var[1:]])) if self.n > 1 \
            else np.array([])
        self.post_c1 = b * np.sqrt(self.acp_prev) / (1.0 - self.acp)
        self.post_c2 = (1.0 - self.acp_prev) * np.sqrt(a) / (1.0 - self.acp)


def respace(base: Tables, kept: List[int]):
    """respace.py:73-87: betas of the process restricted to ``kept`` + the spaced->original map."""
    kept = sorted(kept)
    acp = base.acp[kept]
    prev = np.concatenate([[1.0], acp[:-1]])
    return Tables(1 - acp / prev), kept


def make(kind: str, T: int, respacing: str):
    base = Tables(cosine_betas(T) if kind == "squaredcos_cap_v2" else linear_betas(T))
    if respacing.startswith("ddim"):
        return respace(base, ddim_steps(T, int(respacing[4:])))
    return base, list(range(T))


def _ex(arr: np.ndarray, i: int, like: torch.Tensor) -> torch.Tensor:
    """_extract_into_tensor (gaussian_diffusion.py:880-892): float64 table value -> fp32, broadcast."""
    return torch.full_like(like, float(np.float32(arr[i])))


def _sqrt32(t: torch.Tensor) -> torch.Tensor:
    """IEEE-correct fp32 square root (float64 sqrt, rounded once).  The reference evaluates these
    th.sqrt calls on its CUDA device, where sqrtf is correctly rounded; torch's vectorised CPU sqrt is
    NOT (e.g. sqrt(0.96067816f) returns the lower neighbour 0.98014188 of the correct 0.98014194), so
    the oracle does not use it."""
    return torch.sqrt(t.double()).float()


def predict_xstart(tab: Tables, i: int, x: torch.Tensor, out: torch.Tensor, parameterization: str) -> torch.Tensor:
    if parameterization == "v":       # gaussian_diffusion.py:340-344
        return _ex(tab.sqrt_acp, i, x) * x - _ex(tab.sqrt_1m_acp, i, x) * out
    if parameterization == "eps":     # gaussian_diffusion.py:346-351
        return _ex(tab.sqrt_recip_acp, i, x) * x - _ex(tab.sqrt_recipm1_acp, i, x) * out
    return out.float()


def ddim_step(tab: Tables, i: int, x: torch.Tensor, model_out: torch.Tensor, parameterization: str = "v",
              eta: float = 0.0, clip: bool = False, noise: torch.Tensor = None) -> Dict[str, torch.Tensor]:
    """ddim_sample (gaussian_diffusion.py:531-578) with learned-range variance channels ignored
    (they do not enter the DDIM update)."""
    C = x.shape[-1]
    mo = model_out[..., :C]
    x0 = predict_xstart(tab, i, x, mo, parameterization)
    if clip:
        x0 = x0.clamp(-1, 1)
    eps = (_ex(tab.sqrt_recip_acp, i, x) * x - x0) / _ex(tab.sqrt_recipm1_acp, i, x)
    abar = _ex(tab.acp, i, x)
    abar_prev = _ex(tab.acp_prev, i, x)
    sigma = eta * _sqrt32((1 - abar_prev) / (1 - abar)) * _sqrt32(1 - abar / abar_prev)
    mean_pred = x0 * _sqrt32(abar_prev) + _sqrt32(1 - abar_prev - sigma ** 2) * eps
    if noise is None:
        noise = torch.zeros_like(x)
    mask = 0.0 if i == 0 else 1.0
    return {"sample": mean_pred + mask * sigma * noise, "pred_xstart": x0}


def ancestral_step(tab: Tables, i: int, x: torch.Tensor, model_out: torch.Tensor, noise: torch.Tensor,
                   parameterization: str = "v", clip: bool = False) -> Dict[str, torch.Tensor]:
    """p_sample with LEARNED_RANGE variance (gaussian_diffusion.py:285-293, 329, 394-435)."""
    C = x.shape[-1]
    mo, var = model_out[..., :C], model_out[..., C:]
    min_log = _ex(tab.post_logvar_clipped, i, x)
    max_log = _ex(np.log(tab.betas), i, x)
    frac = (var + 1) / 2                       # in the model-output dtype (fp16 under autocast)
    logvar = frac * max_log + (1 - frac) * min_log
    x0 = predict_xstart(tab, i, x, mo, parameterization)
    if clip:
        x0 = x0.clamp(-1, 1)
    mean = _ex(tab.post_c1, i, x) * x0 + _ex(tab.post_c2, i, x) * x
    mask = 0.0 if i == 0 else 1.0
    return {"sample": mean + mask * torch.exp(0.5 * logvar) * noise, "pred_xstart": x0}


def ddim_loop(model: Callable, x: torch.Tensor, tab: Tables, tmap: List[int], parameterization: str = "v",
              eta: float = 0.0, clip: bool = False, **model_kwargs) -> List[Dict[str, torch.Tensor]]:
    """ddim_sample_loop_progressive (gaussian_diffusion.py:651-698) + _WrappedModel (respace.py:117-129)."""
    outs = []
    for i in range(tab.n - 1, -1, -1):
        t = torch.full((x.shape[0],), tmap[i], dtype=torch.int64)
        out = ddim_step(tab, i, x, model(x, t, **model_kwargs), parameterization, eta, clip)
        outs.append(out)
        x = out["sample"]
    return outs
