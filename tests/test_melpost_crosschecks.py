"""Independent cross-checks of the mel post-processing ORACLE (oracle/edtts_oracle.py, section 8f-3) on CPU.

That section stays labelled "parity unpinned": the reference calls torchaudio (generate_sample.py:115-145), which is not installed
here and cannot be fetched, so no output of the reference exists to pin it.  What CAN be done without torchaudio is to check every
building block against an implementation that does not share code with the oracle:
  * the HTK mel filter bank (norm=None) against transformers.audio_utils.mel_filter_bank (present in the image),
  * the product's fp64 pseudo-inverse (melpost.InverseMelScale) against torch.linalg.lstsq(driver="gelsd") -- what torchaudio calls,
  * torch.stft / torch.istft as the oracle uses them (hann window, center, reflect padding, window-envelope normalisation) against a
    plain numpy transcription of the textbook definitions, through one full Griffin-Lim iteration.
"""
import numpy as np
import torch

from edge_diffusion_tts_amd import CFG
from edge_diffusion_tts_amd import melpost
from oracle import edtts_oracle as O


def test_mel_filter_bank_matches_transformers():
    from transformers.audio_utils import mel_filter_bank
    for (n_fft, n_mels, sr, fmin, fmax) in ((1024, 80, 16000, 0.0, 8000.0), (1024, 80, 22050, 0.0, 8000.0), (400, 40, 16000, 20.0, 7600.0)):
        n_freqs = n_fft // 2 + 1
        ours = O.melscale_fbanks(n_freqs, fmin, fmax, n_mels, sr)
        theirs = torch.from_numpy(mel_filter_bank(n_freqs, n_mels, fmin, fmax, sr, norm=None, mel_scale="htk")).float()
        assert ours.shape == theirs.shape == (n_freqs, n_mels)
        assert float((ours - theirs).abs().max()) < 1e-5  # fp32 arithmetic here (as torchaudio) vs fp64 there; filter values are <= 1
        # and the product's own copy (what the HIP path multiplies with)
        assert torch.equal(melpost.melscale_fbanks(n_freqs, fmin, fmax, n_mels, sr), ours)


def test_pseudo_inverse_equals_gelsd_least_squares():
    cfg = CFG(device="cpu")
    n_freqs = cfg.n_fft // 2 + 1
    inv = melpost.InverseMelScale(n_freqs, cfg.n_mels, cfg.sample_rate, cfg.f_min, cfg.f_max)
    g = torch.Generator().manual_seed(3)
    mel = torch.rand(2, cfg.n_mels, 37, generator=g) * 3.0
    fb = O.melscale_fbanks(n_freqs, cfg.f_min, cfg.f_max, cfg.n_mels, cfg.sample_rate)
    ref = O.inverse_mel_scale(mel, fb)  # relu(lstsq(fb^T, mel, driver="gelsd"))
    ours = torch.relu(torch.matmul(inv.pinv, mel))
    scale = float(ref.abs().max())
    assert float((ours - ref).abs().max()) < 2e-5 * scale
    # fp64 both ways: the two definitions agree to rounding (minimum-norm solution of the rank-deficient system)
    sol64 = torch.linalg.lstsq(fb.t()[None].double(), mel.double(), driver="gelsd").solution
    assert float((torch.matmul(torch.linalg.pinv(fb.t().double()), mel.double()) - sol64).abs().max()) < 1e-9 * scale


def _np_stft(x, n_fft, hop, win):
    """textbook STFT, center=True with reflect padding, periodic hann window, one-sided"""
    pad = n_fft // 2
    xp = np.pad(x, (pad, pad), mode="reflect")
    n_frames = 1 + (len(xp) - n_fft) // hop
    frames = np.stack([xp[i * hop: i * hop + n_fft] * win for i in range(n_frames)], axis=1)
    return np.fft.rfft(frames, axis=0)


def _np_istft(spec, n_fft, hop, win):
    """overlap-add of windowed inverse FFTs, divided by the summed squared window, centre padding removed"""
    n_frames = spec.shape[1]
    frames = np.fft.irfft(spec, n=n_fft, axis=0) * win[:, None]
    out = np.zeros(n_fft + hop * (n_frames - 1))
    env = np.zeros_like(out)
    for i in range(n_frames):
        out[i * hop: i * hop + n_fft] += frames[:, i]
        env[i * hop: i * hop + n_fft] += win ** 2
    pad = n_fft // 2
    return (out / np.where(env > 1e-11, env, 1.0))[pad: len(out) - pad]


def test_one_griffin_lim_iteration_against_numpy_transcription():
    n_fft, hop, T = 256, 64, 24
    g = torch.Generator().manual_seed(5)
    spec = (torch.rand(1, n_fft // 2 + 1, T, generator=g, dtype=torch.float64) * 2.0) ** 2
    ang = torch.complex(torch.rand(1, n_fft // 2 + 1, T, generator=g, dtype=torch.float64), torch.rand(1, n_fft // 2 + 1, T, generator=g, dtype=torch.float64))
    wav = O.griffin_lim(spec, n_fft, hop, n_fft, n_iter=1, angles0=ang)[0].numpy()
    # the same iteration written out with numpy: istft -> stft -> momentum update (tprev = 0) -> unit phases -> final istft
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)  # periodic hann (torch.hann_window default)
    mag = np.sqrt(spec[0].numpy())
    a0 = ang[0].numpy()
    inverse = _np_istft(mag * a0, n_fft, hop, win)
    rebuilt = _np_stft(inverse, n_fft, hop, win)
    angles = rebuilt / (np.abs(rebuilt) + 1e-16)  # momentum term is tprev * mom = 0 in the first iteration
    ref = _np_istft(mag * angles, n_fft, hop, win)
    assert wav.shape == ref.shape == (hop * (T - 1),)
    assert float(np.abs(wav - ref).max()) < 1e-9 * float(np.abs(ref).max() + 1.0)
    # second iteration exercises the momentum term: angles = rebuilt - tprev * 0.99 / 1.99
    wav2 = O.griffin_lim(spec, n_fft, hop, n_fft, n_iter=2, angles0=ang)[0].numpy()
    inverse2 = _np_istft(mag * angles, n_fft, hop, win)
    rebuilt2 = _np_stft(inverse2, n_fft, hop, win)
    a2 = rebuilt2 - rebuilt * (0.99 / 1.99)
    a2 = a2 / (np.abs(a2) + 1e-16)
    ref2 = _np_istft(mag * a2, n_fft, hop, win)
    assert float(np.abs(wav2 - ref2).max()) < 1e-9 * float(np.abs(ref2).max() + 1.0)
