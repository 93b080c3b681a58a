"""DiffusionSchedule -- cosine noise schedule tables + the DDIM / DDPM updates, MI355X path.

API mirror of /root/reference/edge_diffusion_tts/schedule.py:11-266.  The nine public tables are built on
the CPU in fp32 with the same operation sequence as the reference (schedule.py:36-59) so they are bit-equal
to the oracle's, then moved with ``.to(device)``.  ``get_ddim_step`` (schedule.py:157-202) and ``ddpm_step``
(schedule.py:204-238) run as hand-written HIP kernels through the C ABI (include/edtts.h: edtts_ddim_step,
edtts_ddpm_step); there is no CPU fallback for them -- CPU tensors raise.
The light-weight algebra helpers (q_sample, predict_*, get_v_target) are not part of the sampler hot loop
and stay ordinary torch expressions.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch

_TABLES = (
    "betas", "alphas", "alpha_bar", "sqrt_alpha_bar", "sqrt_one_minus_alpha_bar", "sqrt_recip_alpha_bar",
    "sqrt_recip_alpha_bar_minus_one", "posterior_variance", "lambda_t",
)


def _bcast(tab: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    return tab[t].reshape(-1, 1, 1)


class DiffusionSchedule:
    TABLE_NAMES = _TABLES

    def __init__(self, T: int, beta_start: float = 1e-4, beta_end: float = 2e-2, device: str = "cpu"):
        # beta_start / beta_end are accepted and ignored, as in the reference (cosine schedule, s = 0.008).
        self.T = int(T)
        self.device = device
        s = 0.008
        grid = torch.linspace(0, T, T + 1, dtype=torch.float32)
        ac = torch.cos(((grid / T) + s) / (1 + s) * torch.pi * 0.5) ** 2
        ac = ac / ac[0]
        self.betas = torch.clip(1 - (ac[1:] / ac[:-1]), 0.0001, 0.9999)
        self.alphas = 1.0 - self.betas
        self.alpha_bar = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alpha_bar = torch.sqrt(self.alpha_bar)
        self.sqrt_one_minus_alpha_bar = torch.sqrt(1.0 - self.alpha_bar)
        self.sqrt_recip_alpha_bar = torch.sqrt(1.0 / self.alpha_bar)
        self.sqrt_recip_alpha_bar_minus_one = torch.sqrt(1.0 / self.alpha_bar - 1)
        ab_prev = torch.cat([torch.ones(1, dtype=torch.float32), self.alpha_bar[:-1]])
        self.posterior_variance = self.betas * (1.0 - ab_prev) / (1.0 - self.alpha_bar)
        self.lambda_t = torch.log(self.sqrt_alpha_bar / self.sqrt_one_minus_alpha_bar)
        # host copies for the per-step scalars of the fused samplers: no device read (and no sync) at call time, which
        # keeps generate_mel / sample_ddpm capturable into a hipGraph
        self._host = {n: getattr(self, n).numpy().copy() for n in ("alphas", "alpha_bar", "betas", "posterior_variance")}
        self._host_t = {n: getattr(self, n).clone() for n in ("sqrt_alpha_bar", "sqrt_one_minus_alpha_bar", "lambda_t")}  # CPU fp32
        if str(device) != "cpu":
            self.to(device)

    # ------------------------------------------------------------------ tables / movement
    def to(self, device) -> "DiffusionSchedule":
        self.device = device
        for name in _TABLES:
            setattr(self, name, getattr(self, name).to(device))
        return self

    def get_schedule_for_steps(self, num_steps: int) -> List[int]:
        stride = self.T // num_steps
        return list(range(self.T - 1, 0, -stride))[:num_steps]

    # ------------------------------------------------------------------ light algebra (not hot path)
    def q_sample(self, x0, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x0)
        return _bcast(self.sqrt_alpha_bar, t) * x0 + _bcast(self.sqrt_one_minus_alpha_bar, t) * noise, noise

    def predict_x0_from_eps(self, x_t, t, eps):
        return _bcast(self.sqrt_recip_alpha_bar, t) * x_t - _bcast(self.sqrt_recip_alpha_bar_minus_one, t) * eps

    def predict_x0_from_v(self, x_t, t, v):
        return _bcast(self.sqrt_alpha_bar, t) * x_t - _bcast(self.sqrt_one_minus_alpha_bar, t) * v

    def predict_eps_from_v(self, x_t, t, v):
        return _bcast(self.sqrt_one_minus_alpha_bar, t) * x_t + _bcast(self.sqrt_alpha_bar, t) * v

    def get_v_target(self, x0, noise, t):
        return _bcast(self.sqrt_alpha_bar, t) * noise - _bcast(self.sqrt_one_minus_alpha_bar, t) * x0

    # ------------------------------------------------------------------ hot path: HIP kernels
    def get_ddim_step(self, x_t: torch.Tensor, t: torch.Tensor, t_prev: torch.Tensor, eps_pred: torch.Tensor,
                      eta: float = 0.0, noise: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """One DDIM update (x0-prediction, clamp to [-3, 3], deterministic for eta == 0).

        Returns ``(x_prev, x0_pred)`` as fresh tensors.  ``noise`` (only used when eta > 0) is a superset of
        the reference signature: when omitted it is drawn with ``torch.randn_like`` like the reference does.
        """
        from . import native
        if eta > 0 and noise is None:
            noise = torch.randn_like(x_t)
        return native.ddim_step(self.alpha_bar, x_t, t, t_prev, eps_pred, float(eta), noise if eta > 0 else None)

    def ddpm_step(self, x_t: torch.Tensor, t: torch.Tensor, eps_pred: torch.Tensor,
                  noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One ancestral DDPM update; noise defaults to ``torch.randn_like(x_t)`` (reference behaviour)."""
        from . import native
        if noise is None:
            noise = torch.randn_like(x_t)
        return native.ddpm_step(self.alphas, self.alpha_bar, self.betas, self.posterior_variance, x_t, t, eps_pred, noise)

    # ------------------------------------------------------------------ host-side scalars for the fused loop
    def ddim_coefficients(self, t: int, t_prev: int, eta: float = 0.0) -> Tuple[float, float, float, float]:
        """fp32 scalars (sqrt(1-ab), sqrt(ab), sqrt(ab_prev), sqrt(1-ab_prev-sigma^2)) for one batch-uniform DDIM step
        (inference.py:39-42), following the reference's per-element expression sequence (schedule.py:179-196) with
        every operation correctly rounded to fp32: +,-,*,/ on numpy float32 scalars, sqrt through fp64 (exact for
        fp32).  This is what the reference's CPU path produces on a host whose vector sqrt is correctly rounded, and
        it is bit-identical to the in-kernel evaluation of edtts_ddim_step -- unlike torch.sqrt on some hosts."""
        import numpy as np
        f32 = np.float32

        def sqrt32(v):
            return f32(np.sqrt(np.float64(v)))

        tab = self._host["alpha_bar"]
        ab = f32(tab[t])
        ab_prev = f32(tab[t_prev]) if t_prev >= 0 else f32(1.0)
        one = f32(1.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            sigma = f32(eta) * sqrt32(((one - ab_prev) / (one - ab)) * (one - ab / ab_prev))
        return (float(sqrt32(one - ab)), float(sqrt32(ab)), float(sqrt32(ab_prev)),
                float(sqrt32((one - ab_prev) - sigma * sigma)))

    def ddpm_coefficients(self, t: int) -> Tuple[float, float, float]:
        """fp32 scalars (1/sqrt(alpha_t), beta_t/sqrt(1-alpha_bar_t), [t>0]*sqrt(posterior_variance_t)) of one ancestral
        step (schedule.py:227-237), every operation correctly rounded to fp32 (same convention as ddim_coefficients)."""
        import numpy as np
        f32 = np.float32

        def sqrt32(v):
            return f32(np.sqrt(np.float64(v)))

        al = f32(self._host["alphas"][t])
        ab = f32(self._host["alpha_bar"][t])
        be = f32(self._host["betas"][t])
        var = f32(self._host["posterior_variance"][t])
        one = f32(1.0)
        return (float(one / sqrt32(al)), float(be / sqrt32(one - ab)), float((one if t > 0 else f32(0.0)) * sqrt32(var)))


class DPMSolverPP:
    """DPM-Solver++ multistep sampler -- API mirror of /root/reference/edge_diffusion_tts/schedule.py:269-534 (the sampler
    train_v2.validate uses).

    ``sample`` runs as ONE C-ABI call (edtts_sample_multistep) when ``model`` is this package's EdgeDiffusionDecoder: the
    context K/V and all steps' conditioning are built once and each step's x0 conversion, clamp and 1st/2nd/3rd-order
    update are fused into the step's last transformer layer.  The reference's behaviour is kept literally, including
    that the "previous timestep" history holds each step's t_prev and the argument order of the 3rd-order differences.
    The per-step scalar coefficients are evaluated on the host with the reference's own fp32 expressions."""

    def __init__(self, schedule: DiffusionSchedule, order: int = 2, predict_x0: bool = False):
        self.schedule = schedule
        self.order = order
        self.predict_x0 = predict_x0
        self.device = schedule.device

    def to(self, device) -> "DPMSolverPP":
        self.device = device
        self.schedule = self.schedule.to(device)
        return self

    # ---- timestep selection (schedule.py:299-324): equal spacing in log-SNR, nearest table index, clamped to [1, max_t]
    def get_time_steps(self, num_steps: int, max_t: Optional[int] = None) -> torch.Tensor:
        lam_t = self.schedule._host_t["lambda_t"]
        max_t = max_t or (self.schedule.T - 1)
        lam_max, lam_min = lam_t[1].item(), lam_t[max_t].item()
        lams = torch.linspace(lam_min, lam_max, num_steps + 1)
        ts = []
        for lam in lams[:-1]:
            t = int((lam_t - lam).abs().argmin().item())
            ts.append(max(1, min(t, max_t)))
        return torch.tensor(ts, device=self.device, dtype=torch.long)

    # ---- light algebra kept for API compatibility (not used by the fused path)
    def model_to_x0(self, model_output, x_t, t):
        return model_output if self.predict_x0 else self.schedule.predict_x0_from_v(x_t, t, model_output)

    def _tab(self, t):
        s = self.schedule
        return _bcast(s.sqrt_alpha_bar, t), _bcast(s.sqrt_one_minus_alpha_bar, t), _bcast(s.lambda_t, t)

    def first_order_update(self, x, x0_pred, t, t_prev):
        _, sig_t, lam_t = self._tab(t)
        a_p, sig_p, lam_p = self._tab(t_prev)
        h = lam_p - lam_t
        return (sig_p / sig_t) * x + a_p * (1 - torch.exp(-h)) * x0_pred

    def second_order_update(self, x, x0_pred, x0_prev, t, t_prev, t_prev2):
        _, sig_t, lam_t = self._tab(t)
        a_p, sig_p, lam_p = self._tab(t_prev)
        lam_p2 = _bcast(self.schedule.lambda_t, t_prev2)
        h = lam_p - lam_t
        r = (lam_p2 - lam_p) / h
        d1 = (1 / r) * (x0_pred - x0_prev)
        return (sig_p / sig_t) * x + a_p * (1 - torch.exp(-h)) * x0_pred + a_p * ((1 - torch.exp(-h)) / h + 1) * d1 * 0.5

    def third_order_update(self, x, x0_preds, t, t_prev, ts_history):
        _, sig_t, lam_t = self._tab(t)
        a_p, sig_p, lam_p = self._tab(t_prev)
        h = lam_p - lam_t
        d1 = x0_preds[0] - x0_preds[1]
        d2 = x0_preds[0] - 2 * x0_preds[1] + x0_preds[2]
        return ((sig_p / sig_t) * x + a_p * (1 - torch.exp(-h)) * x0_preds[0] + a_p * ((1 - torch.exp(-h)) / h + 1) * d1 * 0.5
                + a_p * ((1 - torch.exp(-h)) / (h ** 2) + 0.5 / h + 0.5) * d2 / 6)

    # ---- per-step scalars of the fused path: {mode, p0, p1, c0, c1, rinv, cB, cC}, reference expressions in fp32 on the host
    def step_coefficients(self, timesteps) -> list:
        a_t, s_t, lam = (self.schedule._host_t[n] for n in ("sqrt_alpha_bar", "sqrt_one_minus_alpha_bar", "lambda_t"))
        ts = [int(v) for v in timesteps]
        out, hist_len, t_hist = [], 0, []
        for i, t in enumerate(ts):
            tp = ts[i + 1] if i < len(ts) - 1 else 0
            h = lam[tp] - lam[t]
            c0 = s_t[tp] / s_t[t]
            c1 = a_t[tp] * (1 - torch.exp(-h))
            p0, p1 = (0.0, 1.0) if self.predict_x0 else (float(a_t[t]), -float(s_t[t]))
            rinv = cB = cC = 0.0
            if self.order == 1 or hist_len == 0:
                mode = 1
            elif self.order == 2 or hist_len == 1:
                mode = 2
                r = (lam[t_hist[-1]] - lam[tp]) / h
                rinv = float(1 / r)
                cB = float(a_t[tp] * ((1 - torch.exp(-h)) / h + 1))
            else:
                mode = 3
                cB = float(a_t[tp] * ((1 - torch.exp(-h)) / h + 1))
                cC = float(a_t[tp] * ((1 - torch.exp(-h)) / (h ** 2) + 0.5 / h + 0.5))
            out.append([float(mode), p0, p1, float(c0), float(c1), rinv, cB, cC])
            hist_len = min(hist_len + 1, 2)
            t_hist = (t_hist + [tp])[-2:]
        return out

    @torch.no_grad()
    def sample(self, model, x_T: torch.Tensor, sem_features: torch.Tensor, num_steps: int = 10, max_t: Optional[int] = None,
               return_intermediates: bool = False, *, sem_idx: Optional[torch.Tensor] = None):
        """x_0 = DPM-Solver++(model, x_T [B,T,n_mels], sem_features [B,S,semantic_dim]) in ``num_steps`` (<= 16) steps.
        ``sem_idx=`` (keyword, a superset of the reference signature) conditions on discrete tokens instead."""
        from . import native
        max_t = max_t or 950
        ts = self.get_time_steps(num_steps, max_t).tolist()
        coefs = self.step_coefficients(ts)
        if sem_features is None and sem_idx is None:
            raise ValueError("Either sem_idx or sem_features must be provided")
        B, T, _ = x_T.shape
        S = sem_features.shape[1] if sem_features is not None else sem_idx.shape[1]
        if len(ts) > model.n_step_emb:
            raise IndexError(f"num_steps={num_steps} exceeds the step embedding table ({model.n_step_emb} rows)")
        packed = model._ensure_packed()
        ws = model.workspace(B, T, S, len(ts), x_T.device)
        x, x0_all = native.sample_multistep(model.dims(), packed, ws, None if sem_features is not None else sem_idx.contiguous(),
                                            None if sem_features is None else sem_features.contiguous(), S,
                                            x_T.to(torch.float32).contiguous(), ts, coefs, return_intermediates)
        if return_intermediates:
            return x, list(x0_all.unbind(0))
        return x
