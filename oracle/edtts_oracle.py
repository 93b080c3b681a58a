"""CPU ORACLE for the DDIM sampler path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional (no nn.Module) restatement, in plain PyTorch CPU tensor algebra, of the reference algorithm
    EdgeInference.generate_mel          /root/reference/edge_diffusion_tts/inference.py:23-53
    EdgeDiffusionDecoder.forward        /root/reference/edge_diffusion_tts/models/decoder.py:66-109
    DiffusionSchedule (+get_ddim_step)  /root/reference/edge_diffusion_tts/schedule.py:26-59,157-238
    DepthwiseSeparableConv.forward      /root/reference/edge_diffusion_tts/layers/conv.py:52-64
written from the math in SURVEY.md section 8a; every function cites the reference lines it follows.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (edge-diffusion-tts_amd/) never does and fails loudly without its HIP library.

Pinning: PINNED.  The reference has no tests or golden vectors of its own, so this
oracle is pinned against outputs of the reference itself, run on CPU in the build container by
tests/golden/make_golden.py (committed, with the fixtures it produced under tests/golden/*.npz) and checked
by tests/test_oracle_vs_golden.py.

All functions take a flat ``sd`` mapping (the reference state-dict key names)
and work in the dtype of the tensors they are given: fp32 is the parity oracle, fp64 (``cast_sd(sd,
torch.float64)``) is the arbiter used for the t=999 amplification band.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ------------------------------------------------------------------------------------------------
# small pieces
# ------------------------------------------------------------------------------------------------
def cast_sd(sd: SD, dtype: torch.dtype) -> SD:
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    y = x @ w.transpose(-1, -2)
    return y if b is None else y + b


def rms_norm(x: Tensor, weight: Tensor, eps: float = 1e-6) -> Tensor:
    """layers/mla.py:46-58 -- x * rsqrt(mean(x^2) + eps) * weight."""
    return x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps) * weight


def ada_rms_norm(x: Tensor, cond: Tensor, sd: SD, prefix: str) -> Tensor:
    """layers/transformer.py:64-68 -- scale is the FIRST half of proj(cond), shift the second.
    With use_adaln=False the block holds a plain RMSNorm under `<prefix>weight` instead (transformer.py:101-104,142-157)."""
    if prefix + "proj.weight" not in sd:
        return rms_norm(x, sd[prefix + "weight"])
    mod = linear(cond, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])
    H = x.shape[-1]
    scale, shift = mod[:, :H], mod[:, H:]
    return rms_norm(x, sd[prefix + "norm.weight"]) * (1 + scale[:, None, :]) + shift[:, None, :]


def gelu_erf(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def time_frequencies(H: int) -> Tensor:
    """layers/embeddings.py:38-41 -- always fp32, divisor (half - 1)."""
    half = H // 2
    return torch.exp(torch.arange(half, dtype=torch.float32) * (-math.log(10000.0) / (half - 1)))


def sinusoidal_time_emb(t: Tensor, H: int, dtype: torch.dtype) -> Tensor:
    """layers/embeddings.py:27-43 -- cat[sin, cos] of t.float() * freqs (fp32 hard-coded in the reference;
    the fp64 arbiter evaluates the trig in fp64 on the same fp32 frequencies)."""
    freqs = time_frequencies(H)
    if dtype == torch.float64:
        args = t.to(torch.float64)[:, None] * freqs.to(torch.float64)[None, :]
    else:
        args = t.float()[:, None] * freqs[None, :]
    return torch.cat([torch.sin(args), torch.cos(args)], dim=1).to(dtype)


def positional_table(max_len: int, H: int) -> Tensor:
    """layers/embeddings.py:119-140 -- interleaved sin/cos table, fp32."""
    pe = torch.zeros(max_len, H)
    pos = torch.arange(0, max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, H, 2) * (-math.log(10000.0) / H))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def band_mask(T: int, window: int) -> Tensor:
    """layers/attention.py:27-30 -- True where |i - j| <= window."""
    idx = torch.arange(T)
    return (idx[None, :] - idx[:, None]).abs() <= window


def attention(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor]) -> Tensor:
    """softmax(q k^T / sqrt(d) [masked]) v on [B, h, T, d] tensors (what F.scaled_dot_product_attention
    computes for layers/attention.py:108-112 and layers/mla.py:175-179, dropout off)."""
    s = (q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    return torch.softmax(s, dim=-1) @ v


def split_heads(x: Tensor, heads: int) -> Tensor:
    B, T, H = x.shape
    return x.reshape(B, T, heads, H // heads).transpose(1, 2)


def merge_heads(x: Tensor) -> Tensor:
    B, h, T, d = x.shape
    return x.transpose(1, 2).reshape(B, T, h * d)


# ------------------------------------------------------------------------------------------------
# decoder pieces (each usable alone for per-op parity tests)
# ------------------------------------------------------------------------------------------------
def time_condition(sd: SD, t: Tensor, step_idx: Optional[Tensor]) -> Tensor:
    """models/decoder.py:26-32,77-80 -- Linear(GELU(Linear(sinus(t)))) + step_emb[step_idx]."""
    w1 = sd["time_emb.1.weight"]
    H = w1.shape[0]
    e = sinusoidal_time_emb(t, H, w1.dtype)
    c = linear(gelu_erf(linear(e, w1, sd["time_emb.1.bias"])), sd["time_emb.3.weight"], sd["time_emb.3.bias"])
    if step_idx is not None:
        c = c + sd["step_emb.weight"][step_idx]
    return c


def context_embed(sd: SD, sem_idx: Optional[Tensor], sem_features: Optional[Tensor]) -> Tensor:
    """models/decoder.py:83-93 -- token_emb gather (or sem_proj of features) + context positional table."""
    if sem_features is not None:
        ctx = linear(sem_features, sd["sem_proj.weight"], sd["sem_proj.bias"])
    elif sem_idx is not None:
        ctx = sd["token_emb.weight"][sem_idx]
    else:
        raise ValueError("Either sem_idx or sem_features must be provided")
    return ctx + sd["context_pos_emb.pe"][: ctx.shape[1]]


def input_embed(sd: SD, x_t: Tensor) -> Tensor:
    """models/decoder.py:96-97 -- in_proj + positional table."""
    return linear(x_t, sd["in_proj.weight"], sd["in_proj.bias"]) + sd["pos_emb.pe"][: x_t.shape[1]]


def self_attention(sd: SD, p: str, x: Tensor, heads: int, window: Optional[int]) -> Tensor:
    """layers/attention.py:77-123 -- qkv rows ordered q|k|v, each head-major; band mask; proj with bias."""
    H = x.shape[-1]
    qkv = linear(x, sd[p + "qkv.weight"])
    q, k, v = (split_heads(qkv[..., i * H:(i + 1) * H], heads) for i in range(3))
    mask = band_mask(x.shape[1], window) if window is not None else None
    return linear(merge_heads(attention(q, k, v, mask)), sd[p + "proj.weight"], sd[p + "proj.bias"])


def cross_kv(sd: SD, p: str, ctx: Tensor, heads: int) -> Tuple[Tensor, Tensor]:
    """layers/mla.py:143-153 -- kv_down -> RMSNorm(rank) -> kv_up; K = first H columns, V = second."""
    c = rms_norm(linear(ctx, sd[p + "kv_down_proj.weight"]), sd[p + "kv_norm.weight"])
    kv = linear(c, sd[p + "kv_up_proj.weight"])
    H = kv.shape[-1] // 2
    return split_heads(kv[..., :H], heads), split_heads(kv[..., H:], heads)


def cross_attention(sd: SD, p: str, x: Tensor, ctx: Tensor, heads: int) -> Tensor:
    """layers/mla.py:118-194 in cross mode -- no RoPE (:158-160), no mask (:164), no biases."""
    q = split_heads(linear(x, sd[p + "q_proj.weight"]), heads)
    k, v = cross_kv(sd, p, ctx, heads)
    return linear(merge_heads(attention(q, k, v, None)), sd[p + "out_proj.weight"])


def feed_forward(sd: SD, p: str, x: Tensor) -> Tensor:
    """layers/transformer.py:13-49 -- up-projection, value = first half, gate = second half, value*silu(gate)."""
    u = linear(x, sd[p + "net.0.weight"], sd[p + "net.0.bias"])
    half = u.shape[-1] // 2
    a, g = u[..., :half], u[..., half:]
    return linear(a * (g * torch.sigmoid(g)), sd[p + "net.3.weight"], sd[p + "net.3.bias"])


def transformer_block(sd: SD, p: str, h: Tensor, ctx: Tensor, cond: Tensor, heads: int, window: Optional[int]) -> Tensor:
    """layers/transformer.py:129-160 -- three pre-norm residual branches."""
    h = h + self_attention(sd, p + "attn.", ada_rms_norm(h, cond, sd, p + "norm1."), heads, window)
    h = h + cross_attention(sd, p + "cross_attn.", rms_norm(h, sd[p + "norm2.weight"]), ctx, heads)
    h = h + feed_forward(sd, p + "ffn.", ada_rms_norm(h, cond, sd, p + "norm3."))
    return h


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = (x - mu).pow(2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def n_layers(sd: SD) -> int:
    return 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))


def decoder_forward(sd: SD, x_t: Tensor, t: Tensor, sem_idx: Optional[Tensor] = None, step_idx: Optional[Tensor] = None,
                    sem_features: Optional[Tensor] = None, *, heads: int = 4, window: Optional[int] = 64) -> Tensor:
    """models/decoder.py:66-109."""
    cond = time_condition(sd, t, step_idx)
    ctx = context_embed(sd, sem_idx, sem_features)
    h = input_embed(sd, x_t)
    for i in range(n_layers(sd)):
        h = transformer_block(sd, f"layers.{i}.", h, ctx, cond, heads, window)
    return linear(layer_norm(h, sd["final_norm.weight"], sd["final_norm.bias"]), sd["out_proj.weight"], sd["out_proj.bias"])


# ------------------------------------------------------------------------------------------------
# schedule
# ------------------------------------------------------------------------------------------------
def schedule_tables(T: int = 1000, dtype: torch.dtype = torch.float32) -> Dict[str, Tensor]:
    """schedule.py:36-59 -- cosine schedule (s = 0.008), betas clipped to [1e-4, 0.9999]; beta_start/end unused."""
    s = 0.008
    g = torch.linspace(0, T, T + 1, dtype=dtype)
    ac = torch.cos(((g / T) + s) / (1 + s) * torch.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = torch.clip(1 - (ac[1:] / ac[:-1]), 0.0001, 0.9999)
    alphas = 1.0 - betas
    ab = torch.cumprod(alphas, dim=0)
    ab_prev = torch.cat([torch.ones(1, dtype=dtype), ab[:-1]])
    sab, s1m = torch.sqrt(ab), torch.sqrt(1.0 - ab)
    return {
        "betas": betas, "alphas": alphas, "alpha_bar": ab, "sqrt_alpha_bar": sab, "sqrt_one_minus_alpha_bar": s1m,
        "sqrt_recip_alpha_bar": torch.sqrt(1.0 / ab), "sqrt_recip_alpha_bar_minus_one": torch.sqrt(1.0 / ab - 1),
        "posterior_variance": betas * (1.0 - ab_prev) / (1.0 - ab), "lambda_t": torch.log(sab / s1m),
    }


def ddim_step(alpha_bar: Tensor, x_t: Tensor, t: Tensor, t_prev: Tensor, eps: Tensor, eta: float = 0.0,
              noise: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """schedule.py:157-202 -- division by sqrt(ab); clamp to [-3, 3]; direction uses the RAW eps."""
    ab = alpha_bar[t][:, None, None]
    ab_prev = torch.where((t_prev >= 0)[:, None, None], alpha_bar[t_prev.clamp(min=0)][:, None, None], torch.ones_like(ab))
    x0 = torch.clamp((x_t - torch.sqrt(1 - ab) * eps) / torch.sqrt(ab), -3, 3)
    sigma = eta * torch.sqrt((1 - ab_prev) / (1 - ab) * (1 - ab / ab_prev))
    direction = torch.sqrt(1 - ab_prev - sigma ** 2) * eps
    x_prev = torch.sqrt(ab_prev) * x0 + direction
    if eta > 0:
        x_prev = x_prev + sigma * noise
    return x_prev, x0


def ddpm_step(tabs: Dict[str, Tensor], x_t: Tensor, t: Tensor, eps: Tensor, noise: Tensor) -> Tensor:
    """schedule.py:204-238 -- mean + [t>0] * sqrt(posterior_variance) * noise."""
    a = tabs["alphas"][t][:, None, None]
    ab = tabs["alpha_bar"][t][:, None, None]
    b = tabs["betas"][t][:, None, None]
    mean = (1.0 / torch.sqrt(a)) * (x_t - (b / torch.sqrt(1.0 - ab)) * eps)
    nz = (t > 0).to(x_t.dtype)[:, None, None]
    return mean + nz * torch.sqrt(tabs["posterior_variance"][t][:, None, None]) * noise


def ddim_timesteps(diff_steps: int, num_steps: int) -> List[Tuple[int, int]]:
    """inference.py:35-41 -- (t, t_prev) pairs."""
    stride = diff_steps // num_steps
    ts = list(range(diff_steps - 1, 0, -stride))[:num_steps]
    return [(t, max(t - stride, 0)) for t in ts]


def generate_mel(sd: SD, alpha_bar: Tensor, sem_idx: Tensor, x_T: Tensor, num_steps: int = 4, diff_steps: int = 1000,
                 *, heads: int = 4, window: Optional[int] = 64, trace: Optional[list] = None) -> Tensor:
    """inference.py:23-53 with the initial noise injected (the reference draws it from the global RNG, F10)."""
    B = sem_idx.shape[0]
    x = x_T
    x0 = None
    for i, (t, t_prev) in enumerate(ddim_timesteps(diff_steps, num_steps)):
        tt = torch.full((B,), t, dtype=torch.long)
        eps = decoder_forward(sd, x, tt, sem_idx, torch.full((B,), i, dtype=torch.long), heads=heads, window=window)
        x, x0 = ddim_step(alpha_bar, x, tt, torch.full((B,), t_prev, dtype=torch.long), eps)
        if trace is not None:
            trace.append({"eps": eps, "x_prev": x, "x0": x0})
    return x0


def dpmpp_timesteps(lambda_t: Tensor, num_steps: int, max_t: Optional[int] = None) -> List[int]:
    """schedule.py:299-324 -- num_steps points equally spaced in log-SNR between lambda[max_t] and lambda[1], each mapped
    to the nearest table index and clamped to [1, max_t]."""
    T = lambda_t.shape[0]
    max_t = max_t or (T - 1)
    lam_max, lam_min = lambda_t[1].item(), lambda_t[max_t].item()
    lams = torch.linspace(lam_min, lam_max, num_steps + 1)
    out = []
    for lam in lams[:-1]:
        t = int((lambda_t - lam).abs().argmin().item())
        out.append(max(1, min(t, max_t)))
    return out


def dpmpp_sample(sd: SD, tabs: Dict[str, Tensor], x_T: Tensor, sem_features: Tensor, num_steps: int = 10, order: int = 2,
                 predict_x0: bool = False, max_t: Optional[int] = None, *, heads: int = 4, window: Optional[int] = 64,
                 intermediates: Optional[list] = None) -> Tensor:
    """schedule.py:440-527 (DPMSolverPP.sample) with its update rules :339-438, restated literally -- including that the
    history of "previous timesteps" holds each step's t_prev (so the 2nd-order ratio r is lambda-wise -1) and that the
    3rd-order differences use [x0, older, newer] in that order."""
    a_t, s_t, lam = tabs["sqrt_alpha_bar"], tabs["sqrt_one_minus_alpha_bar"], tabs["lambda_t"]
    max_t = max_t or 950
    ts = dpmpp_timesteps(lam, num_steps, max_t)
    B = x_T.shape[0]
    x = x_T
    x0_hist: List[Tensor] = []
    t_hist: List[int] = []
    for i, t in enumerate(ts):
        tt = torch.full((B,), t, dtype=torch.long)
        out = decoder_forward(sd, x, tt, None, torch.full((B,), i, dtype=torch.long), sem_features, heads=heads, window=window)
        x0 = out if predict_x0 else a_t[t] * x - s_t[t] * out          # schedule.py:109-125
        x0 = torch.clamp(x0, -3, 3)
        if intermediates is not None:
            intermediates.append(x0)
        tp = ts[i + 1] if i < len(ts) - 1 else 0
        h = lam[tp] - lam[t]
        if order == 1 or len(x0_hist) == 0:
            x = (s_t[tp] / s_t[t]) * x + a_t[tp] * (1 - torch.exp(-h)) * x0
        elif order == 2 or len(x0_hist) == 1:
            h_prev = lam[t_hist[-1]] - lam[tp]
            r = h_prev / h
            d1 = (1 / r) * (x0 - x0_hist[-1])
            x = (s_t[tp] / s_t[t]) * x + a_t[tp] * (1 - torch.exp(-h)) * x0 + a_t[tp] * ((1 - torch.exp(-h)) / h + 1) * d1 * 0.5
        else:
            p = [x0] + x0_hist[-2:]
            d1 = p[0] - p[1]
            d2 = p[0] - 2 * p[1] + p[2]
            x = ((s_t[tp] / s_t[t]) * x + a_t[tp] * (1 - torch.exp(-h)) * p[0] + a_t[tp] * ((1 - torch.exp(-h)) / h + 1) * d1 * 0.5
                 + a_t[tp] * ((1 - torch.exp(-h)) / (h ** 2) + 0.5 / h + 0.5) * d2 / 6)
        x0_hist.append(x0)
        t_hist.append(tp)
        if len(x0_hist) > 2:
            x0_hist.pop(0)
            t_hist.pop(0)
    return x


def sample_ddpm(sd: SD, tabs: Dict[str, Tensor], sem_idx: Tensor, x_T: Tensor, noise_all: Tensor, num_steps: int,
                *, heads: int = 4, window: Optional[int] = 64) -> Tensor:
    """The full-schedule ancestral loop of BASELINE config 5: decoder with step_idx=None (as train.py:155 calls
    it) followed by ddpm_step (schedule.py:204-238), for t = T-1 ... T-num_steps, with the per-step noise injected."""
    T = tabs["alpha_bar"].shape[0]
    B = sem_idx.shape[0]
    x = x_T
    for i in range(num_steps):
        tt = torch.full((B,), T - 1 - i, dtype=torch.long)
        eps = decoder_forward(sd, x, tt, sem_idx, None, heads=heads, window=window)
        x = ddpm_step(tabs, x, tt, eps, noise_all[i])
    return x


# ------------------------------------------------------------------------------------------------
# standalone exported layer named by north_star (not called by the decoder)
# ------------------------------------------------------------------------------------------------
def dsconv_forward(x: Tensor, dw: Tensor, pw: Tensor, pb: Tensor, gn_w: Tensor, gn_b: Tensor, groups: int, eps: float = 1e-5,
                   stride: int = 1) -> Tensor:
    """layers/conv.py:52-64 -- depthwise k-tap conv (stride, zero pad k//2, no bias; conv.py:33-41) -> pointwise 1x1 (+bias) ->
    GroupNorm(groups) -> exact GELU.  x: [B, C_in, T]; dw: [C_in, 1, k]; pw: [C_out, C_in, 1]."""
    B, C, T = x.shape
    ksz = dw.shape[-1]
    pad = ksz // 2
    To = (T + 2 * pad - ksz) // stride + 1
    xp = torch.nn.functional.pad(x, (pad, pad))
    y = torch.zeros(B, C, To, dtype=x.dtype)
    for j in range(ksz):
        y = y + xp[:, :, j:j + (To - 1) * stride + 1:stride] * dw[:, 0, j][None, :, None]
    T = To
    z = torch.einsum("oc,bct->bot", pw[:, :, 0], y) + pb[None, :, None]
    Co = z.shape[1]
    zg = z.reshape(B, groups, (Co // groups) * T)
    mu = zg.mean(dim=-1, keepdim=True)
    var = (zg - mu).pow(2).mean(dim=-1, keepdim=True)
    zn = ((zg - mu) * torch.rsqrt(var + eps)).reshape(B, Co, T)
    return gelu_erf(zn * gn_w[None, :, None] + gn_b[None, :, None])


# ------------------------------------------------------------------------------------------------
# long-form in-painting sampler
# ------------------------------------------------------------------------------------------------
def linspace_times(t_start: int, n: int) -> List[int]:
    """inference_pipeline.py:101-102,165-166 -- torch.linspace(t_start, 0, n + 1).long()[:-1]."""
    return torch.linspace(t_start, 0, n + 1).long()[:-1].tolist()


def inpaint_loop(sd: SD, tabs: Dict[str, Tensor], x_init: Tensor, sem_features: Tensor, times: List[int], step_idx: int,
                 known_mel: Optional[Tensor] = None, overlap_len: int = 0, noise_k: Optional[Tensor] = None, cfg_scale: float = 1.0,
                 *, heads: int = 4, window: Optional[int] = 64) -> Tensor:
    """The loop shared by inference_pipeline.py:97-140 (inpaint_student_sample) and :145-196 (inpaint_teacher_refine):
    per step -- overwrite the first `overlap_len` frames with q_sample(known_mel, t) (:117-123 / :173-176), v-prediction decoder
    call on the sem_features context with a constant step index (:125 / :179), optional classifier-free guidance against an
    all-zero context (:181-185), clamped x0 + eps from v (:127-129 / :187-189), deterministic step to t_next with
    sqrt(alpha_bar[t_next]) / sqrt(1 - alpha_bar[t_next]) (:131-132 / :191-192); finally force the known frames (:135-136 / :194-195).
    noise_k [len(times), B, overlap_len, n_mels]: the torch.randn_like(known_mel) draws of the steps."""
    B = x_init.shape[0]
    x = x_init.clone()
    sab, s1m, ab = tabs["sqrt_alpha_bar"], tabs["sqrt_one_minus_alpha_bar"], tabs["alpha_bar"]
    si = torch.full((B,), step_idx, dtype=torch.long)
    for i, t in enumerate(times):
        t_next = times[i + 1] if i < len(times) - 1 else 0
        tt = torch.full((B,), t, dtype=torch.long)
        if known_mel is not None:
            x[:, :overlap_len, :] = sab[t] * known_mel + s1m[t] * noise_k[i]                     # schedule.py:81-84
        v = decoder_forward(sd, x, tt, None, si, sem_features, heads=heads, window=window)
        if cfg_scale != 1.0:
            vu = decoder_forward(sd, x, tt, None, si, torch.zeros_like(sem_features), heads=heads, window=window)
            v = vu + cfg_scale * (v - vu)
        x0 = torch.clamp(sab[t] * x - s1m[t] * v, -3, 3)                                         # schedule.py:121-125
        eps = s1m[t] * x + sab[t] * v                                                            # schedule.py:138-140
        a_next = ab[t_next]
        x = torch.sqrt(a_next) * x0 + torch.sqrt(1 - a_next) * eps
    if known_mel is not None:
        x[:, :overlap_len, :] = known_mel
    return x


def inpaint_student_sample(sd: SD, tabs, x_init, sem_features, known_mel=None, overlap_len=0, num_steps=4, noise_k=None, **kw) -> Tensor:
    """inference_pipeline.py:97-140: times = linspace(T - 1, 0, num_steps + 1)[:-1], step index 3 ("stage 4")."""
    T = tabs["alpha_bar"].shape[0]
    return inpaint_loop(sd, tabs, x_init, sem_features, linspace_times(T - 1, num_steps), 3, known_mel, overlap_len, noise_k, 1.0, **kw)


def inpaint_teacher_refine(sd: SD, tabs, x_coarse, sem_features, noise, known_mel=None, overlap_len=0, strength=0.2, steps=10,
                           cfg_scale=1.0, noise_k=None, **kw) -> Tensor:
    """inference_pipeline.py:145-196: diffuse x_coarse to t_start = int(T * strength) with `noise` (:160-162), then the loop
    over linspace(t_start, 0, steps + 1)[:-1] with step index 0 and optional CFG."""
    T = tabs["alpha_bar"].shape[0]
    t_start = int(T * strength)
    x = tabs["sqrt_alpha_bar"][t_start] * x_coarse + tabs["sqrt_one_minus_alpha_bar"][t_start] * noise
    return inpaint_loop(sd, tabs, x, sem_features, linspace_times(t_start, steps), 0, known_mel, overlap_len, noise_k, cfg_scale, **kw)


def longform_stitch(sd: SD, tabs, sem_features: Tensor, total_frames: int, chunk_frames: int, overlap_frames: int, chunk_stats,
                    latent_slices, draws, strength: float, steps: int, cfg_scale: float, **kw) -> Tensor:
    """inference_pipeline.py:296-367 -- the sliding-window loop and the stitch: chunk i = inpaint_teacher_refine conditioned on the
    last `overlap_frames` frames of chunk i-1 (:329-346); denormalize_mel with the chunk's (mean, std) (:349-352, utils/audio.py:17-19);
    exp -> LINEAR mel, transposed to [n_mels, frames] (:353-354); accumulated under the trapezoid window of :253-260 at frame
    i * (chunk_frames - overlap_frames) (:357-361); divided by the clamped sum of the windows and trimmed (:364-367).
    draws[i] = {x_coarse, noise, noise_k?}: the chunk's torch.randn draws.  Pinned by tests/golden/longform_stitch.npz (the
    reference's own loop statement run on its own closures)."""
    M = draws[0]["x_coarse"].shape[-1]
    hop = chunk_frames - overlap_frames
    final = torch.zeros(M, total_frames + 1000)
    weights = torch.zeros(1, total_frames + 1000)
    window = torch.ones(1, chunk_frames)
    window[0, :overlap_frames] = torch.linspace(0, 1, overlap_frames)
    window[0, -overlap_frames:] = torch.linspace(1, 0, overlap_frames)
    prev_tail = None
    for i, (l0, l1) in enumerate(latent_slices):
        d = draws[i]
        x = inpaint_teacher_refine(sd, tabs, d["x_coarse"], sem_features[:, l0:l1], d["noise"], prev_tail,
                                   overlap_frames if prev_tail is not None else 0, strength, steps, cfg_scale, d.get("noise_k"), **kw)
        prev_tail = x[:, -overlap_frames:].clone()
        mean, std = chunk_stats[i]
        lin = torch.exp(x * std + mean).transpose(1, 2).squeeze(0)
        final[:, i * hop: i * hop + chunk_frames] += lin * window
        weights[:, i * hop: i * hop + chunk_frames] += window
    return (final / torch.clamp(weights, min=1e-5))[:, :total_frames]


# ------------------------------------------------------------------------------------------------
# mel post-processing: denormalise -> exp -> InverseMelScale -> Griffin-Lim
#
# PARITY UNPINNED for this section: the reference calls torchaudio (generate_sample.py:115-145, inference_pipeline.py:382-396),
# which is not installed in the build container and cannot be fetched, so no golden vector of the reference exists.  What follows
# restates torchaudio's published algorithms (torchaudio 2.x: functional.melscale_fbanks, transforms.InverseMelScale (driver
# "gelsd"), functional.griffinlim) on top of torch.stft / torch.istft -- the same primitives torchaudio itself calls.
# ------------------------------------------------------------------------------------------------
def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk"): triangular filters [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def inverse_mel_scale(melspec: Tensor, fb: Tensor) -> Tensor:
    """torchaudio.transforms.InverseMelScale.forward: relu(lstsq(fb^T, melspec).solution) with the minimum-norm driver "gelsd";
    melspec [B, n_mels, T] -> [B, n_freqs, T]."""
    sol = torch.linalg.lstsq(fb.transpose(-1, -2)[None].to(melspec.dtype), melspec, driver="gelsd").solution
    return torch.relu(sol)


def griffin_lim(specgram: Tensor, n_fft: int, hop: int, win_length: int, n_iter: int, power: float = 2.0, momentum: float = 0.99,
                angles0: Optional[Tensor] = None) -> Tensor:
    """torchaudio.functional.griffinlim (rand_init=True, length=None): specgram [B, n_freqs, T] (power spectrogram) -> waveform
    [B, hop * (T - 1)].  angles0: the complex torch.rand draw (uniform real and imaginary parts) the reference takes first."""
    window = torch.hann_window(win_length, dtype=specgram.dtype)
    mom = momentum / (1 + momentum)
    mag = specgram.pow(1 / power)
    angles = angles0 if angles0 is not None else torch.rand(mag.shape, dtype=torch.complex64 if mag.dtype == torch.float32 else torch.complex128)
    tprev = torch.tensor(0.0, dtype=mag.dtype)
    for _ in range(n_iter):
        inverse = torch.istft(mag * angles, n_fft=n_fft, hop_length=hop, win_length=win_length, window=window)
        rebuilt = torch.stft(inverse, n_fft=n_fft, hop_length=hop, win_length=win_length, window=window, center=True, pad_mode="reflect",
                             normalized=False, onesided=True, return_complex=True)
        angles = rebuilt
        if mom:
            angles = angles - tprev * mom
        angles = angles / (angles.abs() + 1e-16)
        tprev = rebuilt
    return torch.istft(mag * angles, n_fft=n_fft, hop_length=hop, win_length=win_length, window=window)


def mel_to_waveform(mel_n: Tensor, mean: Tensor, std: Tensor, fb: Tensor, n_fft: int, hop: int, win_length: int, n_iter: int = 32,
                    angles0: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """generate_sample.py:115-145: denormalize_mel (utils/audio.py:17-19) -> exp -> transpose -> InverseMelScale -> GriffinLim.
    mel_n [B, T, n_mels]; mean / std [B, 1, n_mels].  Returns (linear power spectrogram [B, n_freqs, T], waveform)."""
    lin_mel = torch.exp(mel_n * std + mean).transpose(1, 2)
    spec = inverse_mel_scale(lin_mel, fb)
    return spec, griffin_lim(spec, n_fft, hop, win_length, n_iter, angles0=angles0)
