"""Deterministic, torch-RNG-independent weights for parity tests and benchmarks.

``synth_state_dict(cfg, seed)`` fills every decoder state-dict tensor from a counter-based integer hash
(splitmix64), so the same weights can be rebuilt bit-for-bit on any machine (the build container that
produced tests/golden/*.npz and the GPU box).  All tensors the reference zero-initialises (out_proj and the
AdaLN projections) get non-zero values, otherwise the decoder output is identically zero.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

_MASK = (1 << 64) - 1


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_MASK)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_MASK)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_MASK)
    return z ^ (z >> np.uint64(31))


def hash_uniform(shape, seed: int, stream: int) -> np.ndarray:
    """float64 uniform in [-1, 1) from splitmix64(seed, stream, element index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([(seed * 0x100000001B3 + stream) & _MASK], dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64) + base
        bits = _splitmix64(idx)
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (2.0 * u - 1.0).reshape(shape)


def sinusoidal_table(max_len: int, dim: int) -> torch.Tensor:
    """Interleaved sin/cos positional table (reference: layers/embeddings.py:119-140), fp32 on CPU."""
    pos = torch.arange(0, max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2) * (-math.log(10000.0) / dim))
    pe = torch.zeros(max_len, dim)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def time_frequencies(dim: int) -> torch.Tensor:
    """fp32 frequency row of the sinusoidal timestep embedding (reference: layers/embeddings.py:38-41)."""
    half = dim // 2
    return torch.exp(torch.arange(half, dtype=torch.float32) * (-math.log(10000.0) / (half - 1)))


def decoder_shapes(cfg, max_pos: int = 1000, max_ctx_pos: int = 512, n_step_emb: int = 16) -> "OrderedDict[str, Tuple[int, ...]]":
    """State-dict key -> shape, in the reference's registration order (models/decoder.py:17-64)."""
    H, M, R = cfg.hidden, cfg.n_mels, cfg.hidden // 2
    F = cfg.hidden * cfg.ffn_mult
    adaln = bool(getattr(cfg, "use_adaln", True))
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["token_emb.weight"] = (cfg.codebook_size, H)
    s["sem_proj.weight"] = (H, cfg.semantic_dim)
    s["sem_proj.bias"] = (H,)
    s["time_emb.1.weight"] = (H, H)
    s["time_emb.1.bias"] = (H,)
    s["time_emb.3.weight"] = (H, H)
    s["time_emb.3.bias"] = (H,)
    s["step_emb.weight"] = (n_step_emb, H)
    s["in_proj.weight"] = (H, M)
    s["in_proj.bias"] = (H,)
    s["pos_emb.pe"] = (max_pos, H)
    s["context_pos_emb.pe"] = (max_ctx_pos, H)
    for i in range(cfg.layers):
        p = f"layers.{i}."
        if adaln:
            s[p + "norm1.norm.weight"] = (H,)
            s[p + "norm1.proj.weight"] = (2 * H, H)
            s[p + "norm1.proj.bias"] = (2 * H,)
        else:  # plain RMSNorm blocks (layers/transformer.py:101-104)
            s[p + "norm1.weight"] = (H,)
        s[p + "attn.qkv.weight"] = (3 * H, H)
        s[p + "attn.proj.weight"] = (H, H)
        s[p + "attn.proj.bias"] = (H,)
        s[p + "norm2.weight"] = (H,)
        s[p + "cross_attn.q_proj.weight"] = (H, H)
        s[p + "cross_attn.kv_down_proj.weight"] = (R, H)
        s[p + "cross_attn.kv_norm.weight"] = (R,)
        s[p + "cross_attn.kv_up_proj.weight"] = (2 * H, R)
        s[p + "cross_attn.out_proj.weight"] = (H, H)
        if adaln:
            s[p + "norm3.norm.weight"] = (H,)
            s[p + "norm3.proj.weight"] = (2 * H, H)
            s[p + "norm3.proj.bias"] = (2 * H,)
        else:  # layers/transformer.py:119-122
            s[p + "norm3.weight"] = (H,)
        s[p + "ffn.net.0.weight"] = (2 * F, H)
        s[p + "ffn.net.0.bias"] = (2 * F,)
        s[p + "ffn.net.3.weight"] = (H, F)
        s[p + "ffn.net.3.bias"] = (H,)
    s["final_norm.weight"] = (H,)
    s["final_norm.bias"] = (H,)
    s["out_proj.weight"] = (M, H)
    s["out_proj.bias"] = (M,)
    return s


def synth_state_dict(cfg, seed: int = 0, max_pos: int = 1000, max_ctx_pos: int = 512) -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = OrderedDict()
    for stream, (key, shape) in enumerate(decoder_shapes(cfg, max_pos, max_ctx_pos).items()):
        if key == "pos_emb.pe":
            out[key] = sinusoidal_table(max_pos, cfg.hidden)
            continue
        if key == "context_pos_emb.pe":
            out[key] = sinusoidal_table(max_ctx_pos, cfg.hidden)
            continue
        u = hash_uniform(shape, seed, stream)
        leaf = key.rsplit(".", 1)[-1]
        if key.endswith("norm.weight") or key.endswith("norm2.weight") or key.endswith("norm1.weight") or key.endswith("norm3.weight") or key == "final_norm.weight" or key.endswith("kv_norm.weight"):
            v = 1.0 + 0.1 * u                      # norm gains near 1
        elif key.endswith("emb.weight"):
            v = 0.5 * u                            # embeddings
        elif ".norm1.proj." in key or ".norm3.proj." in key:
            v = (0.5 / math.sqrt(cfg.hidden)) * u  # AdaLN modulation (zero in the reference init, F4): small but non-zero
        elif leaf == "bias":
            v = 0.05 * u
        else:
            v = u / math.sqrt(shape[-1])           # linear weights ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        out[key] = torch.from_numpy(np.ascontiguousarray(v.astype(np.float32)))
    return out
