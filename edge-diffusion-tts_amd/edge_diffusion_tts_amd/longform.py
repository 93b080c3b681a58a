"""Long-form in-painting sampler -- the sampling core of /root/reference/inference_pipeline.py:97-196,296-367 on MI355X.

The reference keeps ``inpaint_student_sample`` and ``inpaint_teacher_refine`` as closures inside its pipeline script; here they
are methods of :class:`InpaintSampler` with the same arguments and arithmetic (v-prediction decoder on the ``sem_features``
context with a constant step index, q_sample of the previous chunk's tail forced onto the first ``overlap_len`` frames at every
step, optional classifier-free guidance against an all-zero context), each running as ONE C-ABI call
(include/edtts.h: edtts_sample_inpaint): context K/V built once per call (twice with guidance), the blend is a tiny elementwise
kernel, guidance combine + x0 / eps / next-x update are fused into the last transformer layer.  ``generate_long`` is the
reference's chunk loop (sequential: every chunk is conditioned on the tail of the previous one) with its linear cross-fade.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import native
from .schedule import DiffusionSchedule


def linspace_times(t_start: int, n: int) -> List[int]:
    """torch.linspace(t_start, 0, n + 1).long()[:-1] (inference_pipeline.py:101-102,165-166)."""
    return torch.linspace(t_start, 0, n + 1).long()[:-1].tolist()


class InpaintSampler:
    def __init__(self, cfg, schedule: DiffusionSchedule, decoder):
        self.cfg, self.schedule, self.decoder = cfg, schedule, decoder
        self._dev_cache = {}

    def _coefs(self, times: List[int]):
        sab, s1m = self.schedule._host_t["sqrt_alpha_bar"], self.schedule._host_t["sqrt_one_minus_alpha_bar"]
        out = []
        for i, t in enumerate(times):
            t_next = times[i + 1] if i < len(times) - 1 else 0
            a = torch.tensor(float(self.schedule._host["alpha_bar"][t_next]), dtype=torch.float32)
            # sqrt(alpha_next), sqrt(1 - alpha_next) as fp32 tensor ops of the fp32 table value (inference_pipeline.py:131-132)
            out += [float(sab[t]), float(s1m[t]), float(torch.sqrt(a)), float(torch.sqrt(1 - a))]
        return out

    def _run(self, x: torch.Tensor, sem_features: torch.Tensor, times: List[int], step_idx: int, known_mel, overlap_len: int,
             cfg_scale: float, noise_k, seed: int) -> torch.Tensor:
        dec = self.decoder
        B, T, M = x.shape
        S = sem_features.shape[1]
        n = len(times)
        dev = x.device
        x = x.to(torch.float32).contiguous().clone()
        sem_features = sem_features.to(device=dev, dtype=torch.float32).contiguous()
        key = (tuple(times), step_idx, str(dev))
        if key not in self._dev_cache:  # (device copies made once: no H2D copy at call time -> capturable)
            self._dev_cache[key] = (torch.tensor(times, dtype=torch.int64, device=dev),
                                    torch.full((n,), step_idx, dtype=torch.int64, device=dev))
        t_all, s_all = self._dev_cache[key]
        cf = (C.c_float * (4 * n))(*self._coefs(times))
        packed = dec._ensure_packed()
        ws = dec.workspace(B, T, S, n, dev)
        guided = float(cfg_scale) != 1.0
        ws_u = dec.workspace(B, T, S, n, dev, tag="uncond") if guided else None
        zeros = torch.zeros_like(sem_features) if guided else None
        v_u = torch.empty_like(x) if guided else None
        if known_mel is not None:
            known_mel = known_mel.to(device=dev, dtype=torch.float32).contiguous()
            if tuple(known_mel.shape) != (B, overlap_len, M):
                raise ValueError(f"known_mel must be [{B}, {overlap_len}, {M}], got {tuple(known_mel.shape)}")
            if noise_k is not None:
                noise_k = noise_k.to(device=dev, dtype=torch.float32).contiguous()
                if tuple(noise_k.shape) != (n, B, overlap_len, M):
                    raise ValueError(f"noise_k must be [{n}, {B}, {overlap_len}, {M}]")
        else:
            noise_k = None
        p = native._dev_ptr
        native.lib().edtts_sample_inpaint(
            C.byref(dec.dims()), packed.data_ptr(), ws.data_ptr(), None if ws_u is None else ws_u.data_ptr(), B, T, S,
            p(sem_features, torch.float32, "sem_features"), p(zeros, torch.float32, "zeros"), p(x, torch.float32, "x"), n,
            t_all.data_ptr(), s_all.data_ptr(), cf, p(known_mel, torch.float32, "known_mel"), int(overlap_len),
            p(noise_k, torch.float32, "noise_k"), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), float(cfg_scale),
            None if v_u is None else v_u.data_ptr(), native._stream(dev))
        native.check_indices(ws)
        return x

    @torch.no_grad()
    def inpaint_student_sample(self, x_shape, sem_features, known_mel=None, overlap_len: int = 0, num_steps: int = 4, *,
                               x_init: Optional[torch.Tensor] = None, noise_k: Optional[torch.Tensor] = None, seed: int = 0):
        """inference_pipeline.py:97-140.  ``x_init`` / ``noise_k`` inject the draws the reference takes from torch.randn /
        torch.randn_like (parity); otherwise the start noise comes from the library's Philox stream and the per-step q_sample
        noise from the in-kernel generator."""
        dev = sem_features.device
        x = x_init.to(dev) if x_init is not None else native.randn(tuple(x_shape), dev, seed=seed, stream_id=0x51)
        times = linspace_times(self.cfg.diff_steps - 1, num_steps)
        return self._run(x, sem_features, times, 3, known_mel, overlap_len if known_mel is not None else 0, 1.0, noise_k, seed)

    @torch.no_grad()
    def inpaint_teacher_refine(self, x_coarse, sem_features, known_mel=None, overlap_len: int = 0, strength: float = 0.2,
                               steps: int = 10, cfg_scale: float = 1.0, *, noise: Optional[torch.Tensor] = None,
                               noise_k: Optional[torch.Tensor] = None, seed: int = 0):
        """inference_pipeline.py:145-196: q_sample(x_coarse, t_start = int(T * strength)) then ``steps`` guided v-prediction steps."""
        dev = x_coarse.device
        t_start = int(self.cfg.diff_steps * strength)
        if not 0 <= t_start < self.cfg.diff_steps:
            raise IndexError(f"t_start = int(diff_steps * strength) = {t_start} is outside the schedule tables "
                             "(the reference indexes them with it too)")
        nz = noise.to(dev) if noise is not None else native.randn(tuple(x_coarse.shape), dev, seed=seed, stream_id=0x52)
        sab = self.schedule._host_t["sqrt_alpha_bar"][t_start].to(dev)
        s1m = self.schedule._host_t["sqrt_one_minus_alpha_bar"][t_start].to(dev)
        x = sab * x_coarse.to(torch.float32) + s1m * nz  # schedule.q_sample (schedule.py:81-84): plain torch on the device
        times = linspace_times(t_start, steps)
        return self._run(x, sem_features, times, 0, known_mel, overlap_len if known_mel is not None else 0, cfg_scale, noise_k, seed)

    @torch.no_grad()
    def generate_long(self, sem_features: torch.Tensor, total_frames: int, chunk_frames: int, overlap_frames: int,
                      lat_per_frame: float = 0.5, strength: float = 0.999, steps: int = 10, cfg_scale: float = 1.0,
                      seed: int = 0) -> torch.Tensor:
        """The reference's sliding-window loop (inference_pipeline.py:296-361) on normalised mels: chunk i is refined from noise
        with the teacher sampler, conditioned on the last ``overlap_frames`` frames of chunk i-1 (in-painting), and the chunks are
        cross-faded with the reference's linear window.  ``sem_features`` [1, S_total, semantic_dim] are the global semantic
        features; chunk i sees the slice that covers its frames (``lat_per_frame`` latents per mel frame).  Returns
        [1, total_frames, n_mels].  (De-normalisation / exp / vocoding follow in melpost.py.)"""
        dev = sem_features.device
        M = self.cfg.n_mels
        hop = chunk_frames - overlap_frames
        n_chunks = max(1, -(-(total_frames - overlap_frames) // hop))
        final = torch.zeros(1, total_frames + chunk_frames, M, device=dev)
        weights = torch.zeros(1, total_frames + chunk_frames, 1, device=dev)
        window = torch.ones(chunk_frames, device=dev)
        window[:overlap_frames] = torch.linspace(0, 1, overlap_frames, device=dev)
        window[-overlap_frames:] = torch.linspace(1, 0, overlap_frames, device=dev)
        prev_tail = None
        for i in range(n_chunks):
            f0 = i * hop
            l0 = int(f0 * lat_per_frame)
            l1 = max(int((f0 + chunk_frames) * lat_per_frame), l0 + 1)
            z = sem_features[:, l0:l1]
            if z.shape[1] == 0:
                break
            x_coarse = native.randn((1, chunk_frames, M), dev, seed=seed + 2 * i + 1, stream_id=0x53)
            x = self.inpaint_teacher_refine(x_coarse, z.contiguous(), prev_tail, overlap_frames, strength, steps, cfg_scale,
                                            seed=seed + 2 * i)
            prev_tail = x[:, -overlap_frames:].clone()
            final[:, f0:f0 + chunk_frames] += x * window[None, :, None]
            weights[:, f0:f0 + chunk_frames] += window[None, :, None]
        return (final / weights.clamp(min=1e-5))[:, :total_frames]
