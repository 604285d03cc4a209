"""DDIM / ancestral sampling loops over a denoiser callable.

Host-side mirror of the reference's ``GaussianDiffusion`` / ``SpacedDiffusion`` /
``_WrappedModel`` (models/diffusion/gaussian_diffusion.py:145-698,
models/diffusion/respace.py:65-129): same constructor arguments, same method
names and generator protocol (each step yields ``{"sample", "pred_xstart"}``),
same assertions.

What is different by design (MI355X-first):

* The reference re-uploads ~10 one-element tables and materialises ~10
  full-size temporaries per step (``_extract_into_tensor``,
  gaussian_diffusion.py:880-892).  Here every per-step scalar is computed ONCE
  on the host in float32 - following the reference's float32 operation order
  exactly - uploaded as one ``[n_steps, 16]`` table at loop start, and the whole
  update ``x_t, model_out -> x_{t-1}, pred_xstart`` is ONE fused HIP kernel
  (``primx_diffusion_step``) that indexes the table by the step number carried
  as a kernel argument.  No host<->device traffic inside the loop.
* The spaced->original timestep map (respace.py:124-129) lives on the device
  as an int64 vector; step ``i`` hands the model ``map[i].expand(B)``.
* There is no CPU arithmetic path: tensors must live on a HIP device and the
  HIP library must be built, otherwise the loop raises.

Out of scope (no caller on the inference path):
``training_losses``, VLB terms, ``ddim_reverse_sample``, ``cond_fn`` /
``denoised_fn`` hooks (NotImplementedError when passed).
"""
from __future__ import annotations

from typing import Callable, Dict, Iterator, Optional, Sequence

import numpy as np
import torch

from .schedule import (
    DiffusionTables,
    LossType,
    ModelMeanType,
    ModelVarType,
    respaced_betas,
)

COEF_STRIDE = 16  # floats per step in the device coefficient table (see include/primx_hip.h)

# column indices of the coefficient table - keep in sync with csrc/rowops.hip (diffusion_step_kernel)
C_SQRT_ACP, C_SQRT_1M_ACP, C_SQRT_RECIP_ACP, C_SQRT_RECIPM1_ACP = 0, 1, 2, 3
C_POST_MEAN1, C_POST_MEAN2, C_MIN_LOG, C_MAX_LOG, C_FIXED_LOGVAR = 4, 5, 6, 7, 8
C_DDIM_X0, C_DDIM_EPS, C_DDIM_SIGMA, C_NONZERO, C_FIXED_VAR = 9, 10, 11, 12, 13

# PRIMX_PLAN_TIMESTEPS=0: the sampling loops do not announce their timesteps to the model (A/B of DiT.plan_timesteps)
PLAN_TIMESTEPS = __import__("os").environ.get("PRIMX_PLAN_TIMESTEPS", "1") != "0"
# Longest loop that is planned.  Device memory per planned step for DiT-XL: 0.6 MB of 16-bit modulation rows (n x (depth * 9 + 2) * D)
# + 2.1 MB of fp32 u / v rows when the LayerNorm fold applies (2 x 9216 columns x depth; DiT._fold_tables) + their 16-bit A operands
# (0.4 MB): ~0.8 GB at 256 steps, freed when the loop ends (clear_timestep_plan drops the plan).  The fold's share is capped separately:
# loops longer than DiT.fold_max_steps (128: BASELINE configs[3]'s 100 steps fold, 0.33 GB) are planned WITHOUT the fold.
PLAN_MAX_STEPS = 256
_MEAN_CODE = {ModelMeanType.EPSILON: 0, ModelMeanType.START_X: 1, ModelMeanType.VELOCITY: 2}
_VAR_CODE = {
    ModelVarType.FIXED_SMALL: 0,
    ModelVarType.FIXED_LARGE: 1,
    ModelVarType.LEARNED: 2,
    ModelVarType.LEARNED_RANGE: 3,
}


def _f32(a) -> np.ndarray:
    return np.asarray(a, dtype=np.float64).astype(np.float32)


class GaussianDiffusion(DiffusionTables):
    """Sampling half of the reference's ``GaussianDiffusion`` (gaussian_diffusion.py:145-698)."""

    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type):
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        self.loss_type = loss_type
        super().__init__(betas)
        self._dev_cache: Dict = {}

    # ------------------------------------------------------------------ tables
    def step_coefficients(self, eta: float = 0.0) -> np.ndarray:
        """float32 ``[num_timesteps, 16]`` table of every scalar one sampling step needs.

        Each entry reproduces what the reference computes on the fly:
        ``_extract_into_tensor`` = float64 table value cast to float32
        (gaussian_diffusion.py:889); the DDIM scalars follow
        gaussian_diffusion.py:561-573 with every intermediate rounded to float32
        as torch does (``eta`` is a Python float multiplied into a float32 tensor).
        """
        n = self.num_timesteps
        tab = np.zeros((n, COEF_STRIDE), dtype=np.float32)
        tab[:, C_SQRT_ACP] = _f32(self.sqrt_alphas_cumprod)
        tab[:, C_SQRT_1M_ACP] = _f32(self.sqrt_one_minus_alphas_cumprod)
        tab[:, C_SQRT_RECIP_ACP] = _f32(self.sqrt_recip_alphas_cumprod)
        tab[:, C_SQRT_RECIPM1_ACP] = _f32(self.sqrt_recipm1_alphas_cumprod)
        tab[:, C_POST_MEAN1] = _f32(self.posterior_mean_coef1)
        tab[:, C_POST_MEAN2] = _f32(self.posterior_mean_coef2)
        if n > 1:
            tab[:, C_MIN_LOG] = _f32(self.posterior_log_variance_clipped)
        tab[:, C_MAX_LOG] = _f32(np.log(self.betas))
        if self.model_var_type == ModelVarType.FIXED_LARGE and n > 1:
            var = np.append(self.posterior_variance[1], self.betas[1:])
            tab[:, C_FIXED_VAR] = _f32(var)
            tab[:, C_FIXED_LOGVAR] = _f32(np.log(var))
        elif self.model_var_type == ModelVarType.FIXED_SMALL and n > 1:
            tab[:, C_FIXED_VAR] = _f32(self.posterior_variance)
            tab[:, C_FIXED_LOGVAR] = _f32(self.posterior_log_variance_clipped)

        one = np.float32(1.0)
        abar = _f32(self.alphas_cumprod)
        abar_prev = _f32(self.alphas_cumprod_prev)
        eta32 = np.float32(eta)
        with np.errstate(divide="ignore", invalid="ignore"):
            sigma = (eta32 * np.sqrt((one - abar_prev) / (one - abar))) * np.sqrt(one - abar / abar_prev)
            sigma = sigma.astype(np.float32)
            tab[:, C_DDIM_X0] = np.sqrt(abar_prev)
            tab[:, C_DDIM_EPS] = np.sqrt((one - abar_prev) - sigma * sigma)
        tab[:, C_DDIM_SIGMA] = sigma
        tab[:, C_NONZERO] = (np.arange(n) != 0).astype(np.float32)
        return tab

    # ------------------------------------------------------------------ device state
    def _timestep_ids(self) -> Sequence[int]:
        """Values handed to the model for spaced step i (identity for an un-spaced process)."""
        return list(range(self.num_timesteps))

    def _device_state(self, device: torch.device, eta: float):
        key = (str(device), float(eta))
        st = self._dev_cache.get(key)
        if st is None:
            coef = torch.from_numpy(self.step_coefficients(eta)).to(device)
            tmap = torch.tensor(list(self._timestep_ids()), dtype=torch.int64, device=device)
            st = (coef, tmap)
            self._dev_cache = {key: st}  # keep only the latest (eta, device)
        return st

    # ------------------------------------------------------------------ one step
    def _step(self, kind: str, model: Callable, x: torch.Tensor, i: int, **kw):
        from .. import ops  # deferred: importing the sampler must not need the HIP library
        with ops.device_of(x):   # launches go to the current device's stream: make x's device current for the step
            return self._step_on_device(kind, model, x, i, **kw)

    def _step_on_device(self, kind: str, model: Callable, x: torch.Tensor, i: int, *, clip_denoised: bool,
                        model_kwargs: Optional[dict], eta: float, coef: torch.Tensor, tmap: torch.Tensor, planner=None):
        from .. import ops

        if x.dim() != 3:
            raise AssertionError("x must be (B, n_tokens, C)")
        B, nt, C = x.shape
        # (B,) int64 timesteps of the ORIGINAL process (respace.py:124-129): a row of the per-loop [n, B] table (contiguous,
        # so nothing downstream has to copy it), or a broadcast view for the single-step API.  No allocation, no H2D.
        t_model = tmap[i] if tmap.dim() == 2 else tmap[i].expand(B)
        if planner is not None:
            planner.select_planned_timestep(i)        # row i of the table announced in _loop
        try:
            model_output = model(x, t_model, **(model_kwargs or {}))
        finally:
            if planner is not None:
                planner.select_planned_timestep(None)
        if isinstance(model_output, tuple):
            model_output = model_output[0]
        learned = self.model_var_type in (ModelVarType.LEARNED, ModelVarType.LEARNED_RANGE)
        want = (B, nt, C * 2) if learned else (B, nt, C)
        if tuple(model_output.shape) != want:
            raise AssertionError(f"model output shape {tuple(model_output.shape)} != {want}")
        # eta == 0: the reference still draws `th.randn_like(x)` every DDIM step and multiplies it by sigma = 0
        # (gaussian_diffusion.py:569-579).  The sample is identical without the draw; what differs is the process-global
        # RNG position afterwards, so a LATER seeded draw in the same process (the next sample's initial noise when no
        # generator is passed) does not replay a reference run.  Callers that need that pass explicit noise / a generator.
        need_noise = (kind == "ancestral") or (eta != 0.0)
        noise = torch.randn_like(x) if need_noise else None
        sample, pred_xstart = ops.diffusion_step(
            x, model_output, coef, i,
            mean_type=_MEAN_CODE[self.model_mean_type],
            var_type=_VAR_CODE[self.model_var_type],
            ancestral=(kind == "ancestral"),
            clip_denoised=clip_denoised,
            noise=noise,
        )
        return {"sample": sample, "pred_xstart": pred_xstart}

    def _loop(self, kind: str, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs,
              device, progress, eta) -> Iterator[dict]:
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn hooks are outside the accelerated path")
        if self.model_mean_type not in _MEAN_CODE:
            raise NotImplementedError(f"Model Mean type {self.model_mean_type} is not supported!")
        if device is None:
            device = noise.device if noise is not None else next(model.parameters()).device
        device = torch.device(device)
        if not isinstance(shape, (tuple, list)):
            raise AssertionError("shape must be a tuple or list")
        img = noise if noise is not None else torch.randn(*shape, device=device)
        if img.device.type != "cuda":
            raise RuntimeError(
                "the sampler runs on a HIP device only (there is no CPU arithmetic path); "
                f"got a tensor on {img.device}"
            )
        img = img.float().contiguous()
        coef, tmap1 = self._device_state(img.device, eta)
        tmap = tmap1[:, None].expand(-1, img.shape[0]).contiguous()     # [n_steps, B], built once per loop
        # The loop knows every timestep it will ask the model for.  A model that can use that (DiT.plan_timesteps: the
        # timestep-only adaLN modulation of all steps is computed once per loop, several rows per pass over the weights,
        # instead of one row per step) is told; the rows are selected by step index, no device read-back.  `model` is the
        # reference's call convention: a module or a bound method such as `model.forward_with_cfg` (inference.py:306-311).
        owner = getattr(model, "__self__", model)
        planner = owner if (callable(getattr(owner, "plan_timesteps", None)) and PLAN_TIMESTEPS
                            and self.num_timesteps <= PLAN_MAX_STEPS) else None
        if planner is not None:
            planner.plan_timesteps(tmap1)
        indices = range(self.num_timesteps - 1, -1, -1)
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        img0 = img

        def step(x, i):
            with torch.no_grad():   # scoped to the step: a generator must not hold the grad-mode context across yields
                return self._step(kind, model, x, i, clip_denoised=clip_denoised, model_kwargs=model_kwargs, eta=eta, coef=coef,
                                  tmap=tmap, planner=planner)
        try:
            for i in indices:
                out = step(img, i)
                if i == 0:
                    out = self._fold_guard(planner, out, img0, step)     # BEFORE the final yield: consumers stop at the last item
                yield out
                img = out["sample"]
        finally:
            if planner is not None:
                planner.clear_timestep_plan()

    def _fold_guard(self, planner, out: dict, img0: torch.Tensor, step: Callable) -> dict:
        """A model that folded its LayerNorms in fp16 (DiT.fold_ln) has the FINAL sample of the loop checked once (one reduction +
        one read-back per loop; NaN / inf propagate through the diffusion update, clipped or not).  The folded operand is
        normalised with the previous site's statistics, so it leaves the fp16 range only if ONE gated branch multiplies a row's
        spread by > 1e3 - no model of the suite comes near.  If it ever happens the loop is run again with the LayerNorm launches
        (whose operand cannot overflow, as in the reference) and THAT result is returned: same contract as the reference, which
        returns whatever its fp16 arithmetic gives.

        What a caller sees on that (never observed) path: `*_sample_loop` return the second loop's sample.  A consumer of the
        `*_progressive` generators has already been handed the FIRST loop's intermediate items - a generator cannot take them back - and
        gets the second loop's final item last: the trajectory it saw is not one loop's (a RuntimeWarning says so).  The second loop
        draws its own noise where the sampler is stochastic (ancestral steps, DDIM with eta > 0): it is a new sample of the same
        distribution, not a replay.  A NaN that does not come from the fold (bad weights or conditioning) survives the second loop and
        is returned as it is, after the same warning."""
        over = getattr(planner, "fold_overflowed", None) if planner is not None else None
        if not callable(over) or not over(out["sample"]):
            return out
        import warnings
        warnings.warn("non-finite sample after a sampling loop with the fp16 LayerNorm fold: running the loop again with "
                      "LayerNorm launches (PRIMX_DIT_FOLD=0 / model.fold_ln = False avoids the first attempt); items already "
                      "yielded by a *_progressive generator belong to the first loop", RuntimeWarning)
        keep, planner.fold_ln = planner.fold_ln, False
        try:
            img = img0
            for i in range(self.num_timesteps - 1, -1, -1):
                out = step(img, i)
                img = out["sample"]
        finally:
            planner.fold_ln = keep
        return out

    # ------------------------------------------------------------------ public API (reference names)
    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                  cond_fn=None, model_kwargs=None, device=None, progress=False):
        """Ancestral sampling generator (gaussian_diffusion.py:476-529)."""
        return self._loop("ancestral", model, shape, noise, clip_denoised, denoised_fn, cond_fn,
                          model_kwargs, device, progress, 0.0)

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False):
        """gaussian_diffusion.py:437-474."""
        final = None
        for final in self.p_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                    denoised_fn=denoised_fn, cond_fn=cond_fn,
                                                    model_kwargs=model_kwargs, device=device,
                                                    progress=progress):
            pass
        return final["sample"]

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                     cond_fn=None, model_kwargs=None, device=None, progress=False, eta=0.0):
        """DDIM generator (gaussian_diffusion.py:651-698)."""
        return self._loop("ddim", model, shape, noise, clip_denoised, denoised_fn, cond_fn,
                          model_kwargs, device, progress, float(eta))

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0):
        """gaussian_diffusion.py:618-649."""
        final = None
        for final in self.ddim_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                       denoised_fn=denoised_fn, cond_fn=cond_fn,
                                                       model_kwargs=model_kwargs, device=device,
                                                       progress=progress, eta=eta):
            pass
        return final["sample"]

    def _single(self, kind, model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, eta):
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn hooks are outside the accelerated path")
        if tuple(t.shape) != (x.shape[0],):
            raise AssertionError("t must have shape (B,)")
        steps = torch.unique(t).tolist()
        if len(steps) != 1:
            raise NotImplementedError("one fused step handles a single timestep per batch")
        coef, tmap = self._device_state(x.device, eta)
        with torch.no_grad():
            return self._step(kind, model, x.float().contiguous(), int(steps[0]), clip_denoised=clip_denoised,
                              model_kwargs=model_kwargs, eta=eta, coef=coef, tmap=tmap)

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None,
                    model_kwargs=None, eta=0.0):
        """One DDIM step at spaced timestep ``t`` (gaussian_diffusion.py:531-578)."""
        return self._single("ddim", model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, float(eta))

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None):
        """One ancestral step (gaussian_diffusion.py:394-435)."""
        return self._single("ancestral", model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, 0.0)


class SpacedDiffusion(GaussianDiffusion):
    """A process that keeps only ``use_timesteps`` of a base process (respace.py:65-114)."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(kwargs["betas"])
        base = DiffusionTables(kwargs["betas"])
        new_betas, self.timestep_map = respaced_betas(base.alphas_cumprod, self.use_timesteps)
        kwargs["betas"] = new_betas
        super().__init__(**kwargs)

    def _timestep_ids(self):
        return self.timestep_map

    def _scale_timesteps(self, t):
        return t


def create_diffusion(timestep_respacing, noise_schedule="linear", use_kl=False, sigma_small=False,
                     parameterization="eps", learn_sigma=True, rescale_learned_sigmas=False,
                     diffusion_steps=1000) -> SpacedDiffusion:
    """Factory with the reference's signature (models/diffusion/__init__.py:10-52)."""
    from .schedule import get_named_beta_schedule, space_timesteps

    betas = get_named_beta_schedule(noise_schedule, diffusion_steps)
    if use_kl:
        loss_type = LossType.RESCALED_KL
    elif rescale_learned_sigmas:
        loss_type = LossType.RESCALED_MSE
    else:
        loss_type = LossType.MSE
    if timestep_respacing is None or timestep_respacing == "":
        timestep_respacing = [diffusion_steps]
    try:
        mean_type = {"eps": ModelMeanType.EPSILON, "xstart": ModelMeanType.START_X,
                     "v": ModelMeanType.VELOCITY}[parameterization]
    except KeyError:
        raise NotImplementedError("Model Mean Type {} is not supported!".format(parameterization)) from None
    if learn_sigma:
        var_type = ModelVarType.LEARNED_RANGE
    else:
        var_type = ModelVarType.FIXED_SMALL if sigma_small else ModelVarType.FIXED_LARGE
    return SpacedDiffusion(
        use_timesteps=space_timesteps(diffusion_steps, timestep_respacing),
        betas=betas, model_mean_type=mean_type, model_var_type=var_type, loss_type=loss_type,
    )
