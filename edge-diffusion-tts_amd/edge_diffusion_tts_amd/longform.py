"""Long-form in-painting sampler -- the sampling core of /root/reference/inference_pipeline.py:97-196,296-367 on MI355X.

The reference keeps ``inpaint_student_sample`` and ``inpaint_teacher_refine`` as closures inside its pipeline script; here they
are methods of :class:`InpaintSampler` with the same arguments and arithmetic (v-prediction decoder on the ``sem_features``
context with a constant step index, q_sample of the previous chunk's tail forced onto the first ``overlap_len`` frames at every
step, optional classifier-free guidance against an all-zero context), each running as ONE C-ABI call
(include/edtts.h: edtts_sample_inpaint): context K/V built once per call (twice with guidance), the blend is a tiny elementwise
kernel, guidance combine + x0 / eps / next-x update are fused into the last transformer layer.  ``generate_long`` is the
reference's chunk loop (:296-367) as written: per-chunk de-normalisation, exp, cross-fade of LINEAR mels with the trapezoid
window, division by the summed weights (sequential: every chunk is conditioned on the tail of the previous one).
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

    @staticmethod
    def latent_slices(n_chunks: int, hop_samples: int, chunk_samples: int, sample_rate: int) -> List[tuple]:
        """Per-chunk [start_lat, end_lat) into the global semantic features (inference_pipeline.py:308-317): sample position ->
        seconds -> 16 kHz sample -> HuBERT frame (320 samples each), with the reference's float arithmetic and truncations."""
        out = []
        for i in range(n_chunks):
            start_sample = i * hop_samples
            end_sample = start_sample + chunk_samples
            out.append((int(start_sample / sample_rate * 16000) // 320, int(end_sample / sample_rate * 16000) // 320))
        return out

    @staticmethod
    def chunk_plan(total_frames: int, chunk_frames: int, overlap_frames: int, hop_length: int, chunk_samples: Optional[int] = None,
                   overlap_samples: Optional[int] = None, total_samples: Optional[int] = None):
        """(n_chunks, chunk_samples, hop_samples) of the sliding window, inference_pipeline.py:221-225: the sample counts are used
        verbatim when given (the reference fixes THEM and derives the frame counts), else rebuilt as frames * hop_length."""
        chunk_samples = int(chunk_samples) if chunk_samples is not None else chunk_frames * hop_length
        overlap_samples = int(overlap_samples) if overlap_samples is not None else overlap_frames * hop_length
        total_samples = int(total_samples) if total_samples is not None else total_frames * hop_length
        hop_samples = chunk_samples - overlap_samples
        if hop_samples <= 0:
            raise ValueError(f"need overlap_samples < chunk_samples, got {overlap_samples} / {chunk_samples}")
        n_chunks = max(1, -(-(total_samples - overlap_samples) // hop_samples))  # int(np.ceil(...)), :225
        return n_chunks, chunk_samples, hop_samples

    @torch.no_grad()
    def generate_long(self, sem_features: torch.Tensor, total_frames: int, chunk_frames: int, overlap_frames: int,
                      chunk_stats, *, strength: float = 0.999, steps: int = 10, cfg_scale: float = 1.0, seed: int = 0,
                      latent_slices: Optional[List[tuple]] = None, hop_length: Optional[int] = None,
                      sample_rate: Optional[int] = None, draws: Optional[List[dict]] = None,
                      chunk_samples: Optional[int] = None, overlap_samples: Optional[int] = None,
                      total_samples: Optional[int] = None) -> torch.Tensor:
        """The reference's context-aware sliding window (inference_pipeline.py:296-367), statement for statement:

            for chunk i (hop = chunk_frames - overlap_frames frames apart):
                z_q_chunk   = sem_features[:, start_lat:end_lat]                                         (:308-325)
                x_refined   = inpaint_teacher_refine(randn, z_q_chunk, known_mel=prev_mel_tail, overlap_len=overlap_frames, ...)
                prev_tail   = x_refined[:, -overlap_frames:]                                             (:346)
                mel_denorm  = denormalize_mel(x_refined, mean_i, std_i)         per-chunk statistics     (:349-352)
                lin_mel     = exp(mel_denorm)^T                                  LINEAR mel [n_mels, T]   (:353)
                final[:, f0:f0+chunk] += lin_mel * window ;  weights[:, f0:f0+chunk] += window           (:360-361)
            final = (final / clamp(weights, 1e-5))[:, :total_frames]                                     (:364-367)

        with the trapezoid window of :253-260 (linear fade-in over the first and fade-out over the last ``overlap_frames``
        frames).  Returns the stitched LINEAR mel [n_mels, total_frames] -- what the reference hands to its smoothing /
        InverseMelScale / Griffin-Lim tail (melpost.py).

        ``chunk_stats``: one (mean, std) pair per chunk, each broadcastable to [1, 1, n_mels] -- the reference takes them from the
        ground-truth audio of the chunk (normalize_mel of its log-mel, :349-351); that audio front end (torchaudio) is outside the
        path, so the caller supplies the numbers.  ``latent_slices``: the per-chunk [start, end) rows of ``sem_features``; by
        default computed from ``hop_length`` / ``sample_rate`` (cfg values) exactly as the reference does.  ``draws`` (parity
        tests): per chunk a dict with the reference's torch.randn draws ``x_coarse``, ``noise`` and (chunks with a known tail)
        ``noise_k``; otherwise they come from the library's Philox streams.
        ``chunk_samples`` / ``overlap_samples`` / ``total_samples``: the reference works the other way round -- it FIXES the sample
        counts (int(2.0 s * sample_rate), int(0.5 s * sample_rate), wav.shape[1]; :221-225) and derives the frame counts through a
        centred mel transform (frames = samples // hop + 1: 201 / 51 frames for 32000 / 8000 samples at hop 160), so frames * hop
        over-states them (8160 instead of 8000 overlap samples) and the chunk count can come out one short.  A caller that mirrors
        the reference passes its sample counts here and they are used verbatim for the chunk count and the semantic slices; the
        defaults (frames * hop_length) serve callers that think in frames.
        The chunk loop is sequential by construction (chunk i is conditioned on the tail of chunk i-1)."""
        dev = sem_features.device
        M = self.cfg.n_mels
        if not 0 <= overlap_frames < chunk_frames:
            raise ValueError(f"need 0 <= overlap_frames < chunk_frames, got {overlap_frames} / {chunk_frames}")
        hop_frames = chunk_frames - overlap_frames
        hop_length = int(hop_length if hop_length is not None else self.cfg.hop_length)
        sample_rate = int(sample_rate if sample_rate is not None else self.cfg.sample_rate)
        n_chunks, chunk_samples, hop_samples = self.chunk_plan(total_frames, chunk_frames, overlap_frames, hop_length, chunk_samples,
                                                                overlap_samples, total_samples)
        if len(chunk_stats) != n_chunks:
            raise ValueError(f"chunk_stats must hold {n_chunks} (mean, std) pairs, got {len(chunk_stats)}")
        if latent_slices is None:
            latent_slices = self.latent_slices(n_chunks, hop_samples, chunk_samples, sample_rate)
        estimated = total_frames + 1000  # :227 (room for the last, ragged chunk)
        if (n_chunks - 1) * hop_frames + chunk_frames > estimated:
            raise ValueError("chunk geometry exceeds the reference's stitching buffer (total_frames + 1000 frames)")
        final = torch.zeros(M, estimated, device=dev)
        weights = torch.zeros(1, estimated, device=dev)
        window = torch.ones(1, chunk_frames, device=dev)
        if overlap_frames > 0:  # (:253-260; with no overlap the window is flat and no tail is handed on)
            window[0, :overlap_frames] = torch.linspace(0, 1, overlap_frames, device=dev)
            window[0, -overlap_frames:] = torch.linspace(1, 0, overlap_frames, device=dev)
        prev_tail = None
        for i in range(n_chunks):
            l0, l1 = latent_slices[i]
            z = sem_features[:, l0:l1].contiguous()
            if z.shape[1] == 0:
                raise ValueError(f"chunk {i}: empty semantic slice [{l0}:{l1}] of {sem_features.shape[1]} rows")
            d = draws[i] if draws is not None else {}
            x_coarse = d["x_coarse"].to(dev) if "x_coarse" in d else native.randn((1, chunk_frames, M), dev, seed=seed + 2 * i + 1, stream_id=0x53)
            x = self.inpaint_teacher_refine(x_coarse, z, prev_tail, overlap_frames if prev_tail is not None else 0, strength, steps,
                                            cfg_scale, noise=d.get("noise"), noise_k=d.get("noise_k"), seed=seed + 2 * i)
            prev_tail = x[:, -overlap_frames:].clone() if overlap_frames > 0 else None
            mean, std = chunk_stats[i]
            mean = torch.as_tensor(mean, dtype=torch.float32, device=dev)
            std = torch.as_tensor(std, dtype=torch.float32, device=dev)
            lin = torch.exp(x * std + mean).transpose(1, 2).squeeze(0)  # utils/audio.py:17-19, then :353-354
            f0 = i * hop_frames
            final[:, f0:f0 + chunk_frames] += lin * window
            weights[:, f0:f0 + chunk_frames] += window
        return (final / torch.clamp(weights, min=1e-5))[:, :total_frames]
