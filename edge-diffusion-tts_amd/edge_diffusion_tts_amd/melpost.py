"""Mel post-processing on MI355X -- what the reference's scripts do right after the sampler (generate_sample.py:115-145,
inference_pipeline.py:382-396): ``denormalize_mel`` (utils/audio.py:17-19) -> ``exp`` -> ``torchaudio.transforms.InverseMelScale``
-> ``torchaudio.transforms.GriffinLim``.  The two transform classes keep torchaudio's constructor arguments and call
conventions so that the reference's scripts can swap the import; their arithmetic runs in libedtts_hip.so
(include/edtts.h: edtts_mel_to_spec, edtts_griffin_lim).  The constant tables (mel filter bank and its pseudo-inverse, Hann
window, FFT twiddles) are built once on the host, like the schedule tables.

PARITY UNPINNED: torchaudio is not available offline, so these ops are checked against the oracle's restatement of torchaudio's
published algorithm (the test suite's CPU oracle), not against outputs of the reference itself.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import native


def normalize_mel(mel: torch.Tensor):
    """utils/audio.py:10-14 (plain tensor algebra; statistics only, not on the hot path)."""
    mean = mel.mean(dim=1, keepdim=True)
    std = mel.std(dim=1, keepdim=True).clamp_min(1e-5)
    return (mel - mean) / std, mean, std


def denormalize_mel(mel_n: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """utils/audio.py:17-19."""
    return mel_n * std + mean


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") -> [n_freqs, n_mels] (CPU, fp32)."""
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


class InverseMelScale(torch.nn.Module):
    """torchaudio.transforms.InverseMelScale(n_stft, n_mels, sample_rate, f_min, f_max, norm=None, mel_scale="htk", driver="gelsd"):
    melspec [..., n_mels, T] -> relu(least-squares spectrogram) [..., n_stft, T]."""

    def __init__(self, n_stft: int, n_mels: int = 128, sample_rate: int = 16000, f_min: float = 0.0, f_max: Optional[float] = None,
                 norm: Optional[str] = None, mel_scale: str = "htk", driver: str = "gelsd"):
        super().__init__()
        if norm is not None or mel_scale != "htk" or driver != "gelsd":
            raise NotImplementedError("only norm=None, mel_scale='htk', driver='gelsd' (what the reference uses) are built")
        f_max = float(sample_rate // 2) if f_max is None else f_max
        fb = melscale_fbanks(n_stft, f_min, f_max, n_mels, sample_rate)
        self.n_stft, self.n_mels = n_stft, n_mels
        self.register_buffer("fb", fb)
        # minimum-norm least squares == multiplication by the pseudo-inverse (evaluated once, in fp64)
        self.register_buffer("pinv", torch.linalg.pinv(fb.t().double()).float().contiguous())

    @torch.no_grad()
    def forward(self, melspec: torch.Tensor, *, log_normalized: Optional[tuple] = None) -> torch.Tensor:
        shape = melspec.shape
        x = melspec.reshape(-1, shape[-2], shape[-1]).transpose(1, 2).contiguous()  # [B, T, n_mels]: the kernels' frame-major layout
        return self._spec(x, None, None).reshape(shape[:-2] + (self.n_stft, shape[-1]))

    def _spec(self, mel_btm: torch.Tensor, mean, std) -> torch.Tensor:
        B, T, M = mel_btm.shape
        if M != self.n_mels:
            raise ValueError(f"expected {self.n_mels} mel bins, got {M}")
        p = native._dev_ptr
        spec = torch.empty(B, self.n_stft, T, dtype=torch.float32, device=mel_btm.device)
        native.lib().edtts_mel_to_spec(p(mel_btm, torch.float32, "mel"), p(mean, torch.float32, "mean"), p(std, torch.float32, "std"),
                                       p(self.pinv, torch.float32, "pinv"), B, T, M, self.n_stft, spec.data_ptr(), native._stream(mel_btm.device))
        return spec

    @torch.no_grad()
    def from_normalized(self, mel_n: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
        """Fused generate_sample.py:115-144: denormalize_mel -> exp -> (transpose) -> inverse mel scale; mel_n [B, T, n_mels],
        mean / std [B, 1, n_mels] -> power spectrogram [B, n_stft, T]."""
        B, T, M = mel_n.shape
        return self._spec(mel_n.contiguous(), mean.expand(B, 1, M).reshape(B, M).contiguous(), std.expand(B, 1, M).reshape(B, M).contiguous())


class GriffinLim(torch.nn.Module):
    """torchaudio.transforms.GriffinLim(n_fft, n_iter=32, win_length=None, hop_length=None, power=2.0, momentum=0.99, length=None,
    rand_init=True): specgram [..., n_fft // 2 + 1, T] -> waveform [..., hop * (T - 1)]."""

    def __init__(self, n_fft: int = 400, n_iter: int = 32, win_length: Optional[int] = None, hop_length: Optional[int] = None,
                 power: float = 2.0, momentum: float = 0.99, length: Optional[int] = None, rand_init: bool = True):
        super().__init__()
        win_length = n_fft if win_length is None else win_length
        if win_length != n_fft or length is not None or not rand_init:
            raise NotImplementedError("win_length == n_fft, length=None, rand_init=True (what the reference uses) are built")
        if not 0 <= momentum < 1:
            raise ValueError("momentum must be in [0, 1)")
        self.n_fft, self.n_iter, self.hop = n_fft, n_iter, (win_length // 2 if hop_length is None else hop_length)
        self.power, self.momentum = power, momentum
        self.register_buffer("window", torch.hann_window(win_length))
        q = torch.arange(n_fft // 2, dtype=torch.float64) * (-2.0 * math.pi / n_fft)
        self.register_buffer("twiddle", torch.stack([torch.cos(q), torch.sin(q)], dim=1).float().contiguous())

    @torch.no_grad()
    def forward(self, specgram: torch.Tensor, *, angles0: Optional[torch.Tensor] = None, seed: int = 0) -> torch.Tensor:
        shape = specgram.shape
        spec = specgram.reshape(-1, shape[-2], shape[-1]).to(torch.float32).contiguous()
        B, F, T = spec.shape
        if F != self.n_fft // 2 + 1:
            raise ValueError(f"expected {self.n_fft // 2 + 1} frequency bins, got {F}")
        n = C.c_size_t(0)
        native.lib().edtts_griffin_lim_scratch_floats(B, T, self.n_fft, self.hop, C.byref(n))
        scratch = torch.empty(n.value, dtype=torch.float32, device=spec.device)
        wave = torch.empty(B, self.hop * (T - 1), dtype=torch.float32, device=spec.device)
        a0 = None
        if angles0 is not None:
            a0 = torch.view_as_real(angles0.reshape(B, F, T).to(torch.complex64)).contiguous()
        p = native._dev_ptr
        native.lib().edtts_griffin_lim(p(spec, torch.float32, "specgram"), B, T, self.n_fft, self.hop, p(self.window, torch.float32, "window"),
                                       p(self.twiddle, torch.float32, "twiddle"), self.n_iter, float(self.momentum), float(self.power),
                                       p(a0, torch.float32, "angles0"), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), scratch.data_ptr(),
                                       wave.data_ptr(), native._stream(spec.device))
        return wave.reshape(shape[:-2] + (wave.shape[-1],))


class MelVocoder(torch.nn.Module):
    """generate_sample.py:115-145 in one object: normalised mel [B, T, n_mels] (+ the utterance's mean / std) -> waveform."""

    def __init__(self, cfg, n_iter: int = 32):
        super().__init__()
        self.inverse_mel = InverseMelScale(n_stft=cfg.n_fft // 2 + 1, n_mels=cfg.n_mels, sample_rate=cfg.sample_rate, f_min=cfg.f_min, f_max=cfg.f_max)
        self.griffin_lim = GriffinLim(n_fft=cfg.n_fft, n_iter=n_iter, win_length=cfg.win_length, hop_length=cfg.hop_length, power=2.0)

    @torch.no_grad()
    def forward(self, mel_n: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, *, angles0=None, seed: int = 0):
        spec = self.inverse_mel.from_normalized(mel_n, mean, std)
        return self.griffin_lim(spec, angles0=angles0, seed=seed)
