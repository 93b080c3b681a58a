"""EdgeInference -- API mirror of /root/reference/edge_diffusion_tts/inference.py:12-62 on MI355X.

``generate_mel`` reproduces the reference's few-step DDIM loop literally (timestep list, step indices, eps
interpretation of the decoder output, x0 of the last step returned) but runs it as ONE C-ABI call
(edtts_generate): conditioning rows for all steps and the cross-attention K/V cache are computed once, every
transformer layer is one fused kernel, and each step's last layer fuses final_norm + out_proj + the DDIM update.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import native
from .config import CFG
from .schedule import DiffusionSchedule


class EdgeInference:
    def __init__(self, cfg: CFG, schedule: DiffusionSchedule, encoder, decoder):
        self.cfg = cfg
        self.schedule = schedule
        self.encoder = encoder
        self.decoder = decoder
        self.device = cfg.device
        self._t_cache = {}

    @torch.no_grad()
    def generate_mel(self, sem_idx: torch.Tensor, num_steps: int = 4, temperature: float = 1.0, *,
                     x_T: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
                     seed: Optional[int] = None, batch_offset: int = 0) -> torch.Tensor:
        """Mel [B, 2*S, n_mels] from semantic tokens [B, S] with ``num_steps`` DDIM steps (1..16).

        ``x_T`` / ``generator`` / ``seed`` extend the reference signature: the reference draws the start noise from the global
        device RNG (inference.py:33), which cannot match a CPU run; parity tests inject the oracle's noise instead.  With
        ``seed`` the noise comes from the library's counter-based Philox stream at global row ``batch_offset`` (edtts_randn):
        a rank holding rows [lo, hi) of a larger batch passes ``batch_offset=lo`` and draws exactly what one GPU would have
        drawn for those rows.
        """
        if self.encoder is not None and hasattr(self.encoder, "eval"):
            self.encoder.eval()
        self.decoder.eval()
        B, S = sem_idx.shape[0], sem_idx.shape[1]
        T_out = 2 * S
        dev = sem_idx.device if sem_idx.is_cuda else torch.device(self.device)
        sem_idx = sem_idx.to(dev)
        if x_T is None and seed is not None:
            x_T = native.randn((B, T_out, self.cfg.n_mels), dev, seed, 0, int(batch_offset) * T_out * self.cfg.n_mels, temperature)
        elif x_T is None:
            x_T = torch.randn(B, T_out, self.cfg.n_mels, device=dev, generator=generator) * temperature
        elif tuple(x_T.shape) != (B, T_out, self.cfg.n_mels):
            raise ValueError(f"x_T must be [{B}, {T_out}, {self.cfg.n_mels}], got {tuple(x_T.shape)}")
        x_T = x_T.to(device=dev, dtype=torch.float32).contiguous()

        stride = self.cfg.diff_steps // num_steps
        timesteps = list(range(self.cfg.diff_steps - 1, 0, -stride))[:num_steps]
        if len(timesteps) > self.decoder.n_step_emb:
            raise IndexError(f"num_steps={num_steps} exceeds the step embedding table ({self.decoder.n_step_emb} rows)")
        coefs = [self.schedule.ddim_coefficients(t, max(t - stride, 0), eta=0.0) for t in timesteps]

        packed = self.decoder._ensure_packed()
        ws = self.decoder.workspace(B, T_out, S, len(timesteps), dev)
        return native.generate(self.decoder.dims(), packed, ws, sem_idx.contiguous(), x_T, timesteps, coefs)

    @torch.no_grad()
    def sample_ddpm(self, sem_idx: torch.Tensor, num_steps: Optional[int] = None, temperature: float = 1.0, *,
                    x_T: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None, seed: int = 0,
                    generator: Optional[torch.Generator] = None, batch_offset: int = 0) -> torch.Tensor:
        """Full-schedule ancestral (DDPM) sampler: for t = T-1 ... T-num_steps: eps = decoder(x, t, sem_idx) (step_idx=None),
        x = schedule.ddpm_step(x, t, eps).  This is the loop BASELINE config 5 names; the reference never wrote it (its
        generate_mel cannot exceed 16 steps) but ships both pieces (train.py:155, schedule.py:204-238).
        One C-ABI call: conditioning rows of all steps and the cross-attention cache are built once, the DDPM update is fused
        into every step's last layer.  ``noise`` [num_steps, B, 2S, n_mels] injects the per-step draws (parity); otherwise an
        in-kernel Philox generator keyed by (seed, step, global element) supplies them, where ``batch_offset`` is the global index
        of this call's first utterance (a rank sampling rows [lo, hi) of a batch passes lo: shard-count-invariant draws).
        Graph-capturable; seed and step are baked into a captured graph, so a replay repeats the same noise."""
        B, S = sem_idx.shape
        T_out = 2 * S
        dev = sem_idx.device if sem_idx.is_cuda else torch.device(self.device)
        sem_idx = sem_idx.to(dev).contiguous()
        n = self.cfg.diff_steps if num_steps is None else int(num_steps)
        if not 1 <= n <= self.cfg.diff_steps:
            raise ValueError(f"num_steps must be in [1, {self.cfg.diff_steps}]")
        if x_T is None:
            x_T = torch.randn(B, T_out, self.cfg.n_mels, device=dev, generator=generator) * temperature
        x_T = x_T.to(device=dev, dtype=torch.float32).contiguous()
        ts = list(range(self.cfg.diff_steps - 1, self.cfg.diff_steps - 1 - n, -1))
        key = (n, str(dev))
        if key not in self._t_cache:  # device copy made once (an H2D copy is not allowed inside graph capture)
            self._t_cache[key] = torch.tensor(ts, dtype=torch.int64, device=dev)
        t_all = self._t_cache[key]
        coefs = [self.schedule.ddpm_coefficients(t) for t in ts]
        if noise is not None:
            if tuple(noise.shape) != (n, B, T_out, self.cfg.n_mels):
                raise ValueError(f"noise must be [{n}, {B}, {T_out}, {self.cfg.n_mels}]")
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
        packed = self.decoder._ensure_packed()
        ws = self.decoder.workspace(B, T_out, S, n, dev)
        return native.sample_ddpm(self.decoder.dims(), packed, ws, sem_idx, x_T, t_all, coefs, noise, seed, batch_offset)

    # alias some callers may expect from the task description; not part of the reference API
    generate = generate_mel

    @torch.no_grad()
    def generate_from_audio(self, wav: torch.Tensor, num_steps: int = 4) -> torch.Tensor:
        """wav -> encoder -> generate_mel (inference.py:55-62).  The encoder (HuBERT + quantiser) is outside this
        package; any callable returning the reference's 5-tuple with sem_idx second works."""
        if wav.dim() == 1:
            wav = wav.unsqueeze(0)
        wav = wav.to(self.device)
        _, sem_idx, _, _, _ = self.encoder(wav)
        return self.generate_mel(sem_idx, num_steps)
