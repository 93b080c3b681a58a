#!/usr/bin/env python3
"""Round-3 additions to tests/golden/*.npz, made by running the REFERENCE on CPU (build container only).

    python tests/golden/make_golden_r3.py [name ...]

Fixtures (inputs + the reference's outputs; no reference source is stored):
  longform_stitch   the sliding-window loop of inference_pipeline.py:296-361 and the weight normalisation of :364-367, run AS THEY
                    ARE: the `for` statement and the three statements after it are taken from the AST of inference_pipeline.main()
                    at generation time and executed with the reference's own closures (inpaint_teacher_refine), its own
                    normalize_mel / denormalize_mel (edge_diffusion_tts/utils/audio.py) and this repo's synthetic weights.  What is
                    NOT available offline is stood in for: `wav` is synthetic, `mel_transform` (torchaudio MelSpectrogram) is a
                    deterministic positive function of the chunk, `z_q_global` (HuBERT features) is synthetic.  Every torch.randn /
                    randn_like draw of the run and the per-chunk (mean, std) the loop derives are recorded, so that the build's
                    generate_long can be driven with exactly the same numbers.
  forward_ffn_mult  decoder forwards at ffn_mult = 4 and 1 (the reference default is 2)
  dsconv_stride     DepthwiseSeparableConv(stride = 2, 3) (layers/conv.py:25-64), and shapes outside the fused kernel's class
  bf16_sampler      the reference's own generate_mel under torch.autocast("cpu", bfloat16) next to its fp32 run (4-step DDIM, hidden
                    64 / 2 heads of 32 / 2 layers, the smallest bf16-capable shape): the error distribution a bf16 implementation of
                    this sampler has by the reference's own standard.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402,F401  (sets sys.path for the reference + this repo's synth module, chdirs to a scratch dir)
from make_golden import OUT, make_decoder, npf, ref, rnd, rnd_idx  # noqa: E402
from make_golden_r2 import _extract_pipeline_closures  # noqa: E402


class _RecordingTorch:
    """Stands in for the `torch` module inside the extracted code: everything is torch's, but randn / randn_like are recorded."""

    def __init__(self):
        self.draws = []

    def __getattr__(self, name):
        return getattr(torch, name)

    def randn(self, *a, **k):
        k.pop("device", None)
        v = torch.randn(*a, **k)
        self.draws.append(("randn", v))
        return v

    def randn_like(self, x, **k):
        v = torch.randn_like(x)
        self.draws.append(("randn_like", v))
        return v


def _extract_stitch_loop():
    """The `for i in tqdm.tqdm(range(num_chunks))` statement of inference_pipeline.main() and the statements that normalise and
    trim the stitched mel right after it (:364-367), compiled on their own.  Read at generation time, never stored."""
    import ast
    tree = ast.parse(open("/root/reference/inference_pipeline.py").read())
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    idx = next(i for i, n in enumerate(main.body) if isinstance(n, ast.For) and "num_chunks" in ast.unparse(n.iter))
    tail = []
    for n in main.body[idx + 1:]:
        src = ast.unparse(n)
        if src.startswith("print("):
            break
        tail.append(n)
    assert [ast.unparse(n).split("=")[0].strip() for n in tail] == ["final_weights", "final_mel", "final_mel"], [ast.unparse(n) for n in tail]
    return compile(ast.Module(body=[main.body[idx]] + tail, type_ignores=[]), "/root/reference/inference_pipeline.py", "exec")


@torch.no_grad()
def longform_stitch():
    import torch.nn.functional as F
    from edge_diffusion_tts.utils.audio import denormalize_mel, normalize_mel
    cfg = ref.CFG(device="cpu")
    dec = make_decoder(cfg, seed=0)
    sch = ref.DiffusionSchedule(cfg.diff_steps)
    rt = _RecordingTorch()
    ns = {"torch": rt, "cfg": cfg, "device": "cpu", "schedule": sch, "student_decoder": dec, "teacher_decoder": dec}
    exec(_extract_pipeline_closures(), ns)

    # geometry: hop_length 256 (CFG default) -> a "chunk" of 48 mel frames, 12 frames of overlap, 3 chunks, the last one ragged
    hop_len = cfg.hop_length
    chunk_frames, overlap_frames = 48, 12
    chunk_samples, overlap_samples = chunk_frames * hop_len, overlap_frames * hop_len
    hop_samples = chunk_samples - overlap_samples
    total_frames = 100
    total_samples = total_frames * hop_len
    num_chunks = int(np.ceil((total_samples - overlap_samples) / hop_samples))
    assert num_chunks == 3
    wav = rnd((1, total_samples), 81, 0, 0.3)
    stats = []

    def mel_transform(w):
        """stand-in for torchaudio's MelSpectrogram (absent offline): [1, n] -> positive [1, 80, n // hop_length]; a chunk shorter
        than chunk_samples (the last one) gives fewer frames, exactly as the real transform would"""
        n = w.shape[1] // hop_len
        fr = w[:, : n * hop_len].reshape(1, n, hop_len)
        e = fr.pow(2).mean(dim=2)  # [1, n] frame energy
        k = torch.arange(cfg.n_mels, dtype=torch.float32)[None, :, None]
        return (e[:, None, :] * (1.0 + 0.5 * torch.cos(0.37 * k + 3.0 * fr.mean(dim=2)[:, None, :])) * torch.exp(-0.03 * k) + 1e-4)

    def normalize_rec(m):
        out = normalize_mel(m)
        stats.append((out[1].clone(), out[2].clone()))
        return out

    class _Tqdm:
        @staticmethod
        def tqdm(it):
            return it

    z_q_global = rnd((1, 80, cfg.semantic_dim), 81, 1)  # 16 kHz / 320 = 50 latents per second; 100 frames * 256 / 22050 s = 1.16 s
    estimated_frames = total_frames + 1000
    window_mask = torch.ones(1, chunk_frames)
    window_mask[0, :overlap_frames] = torch.linspace(0, 1, overlap_frames).unsqueeze(0)  # inference_pipeline.py:253-260, arithmetic restated
    window_mask[0, -overlap_frames:] = torch.linspace(1, 0, overlap_frames).unsqueeze(0)
    steps, strength, scale = 5, 0.6, 1.5
    ns.update(dict(tqdm=_Tqdm, F=F, num_chunks=num_chunks, hop_samples=hop_samples, chunk_samples=chunk_samples, wav=wav,
                   mel_transform=mel_transform, z_q_global=z_q_global, chunk_frames=chunk_frames, overlap_frames=overlap_frames,
                   refine_strength=strength, refine_steps=steps, cfg_scale=scale, normalize_mel=normalize_rec,
                   denormalize_mel=denormalize_mel, hop_frames=chunk_frames - overlap_frames,
                   final_mel=torch.zeros(cfg.n_mels, estimated_frames), final_weights=torch.zeros(1, estimated_frames),
                   window_mask=window_mask, total_frames=total_frames, prev_mel_tail=None))
    torch.manual_seed(17)
    exec(_extract_stitch_loop(), ns)
    final = ns["final_mel"]
    assert tuple(final.shape) == (cfg.n_mels, total_frames) and len(stats) == num_chunks
    # the draws, chunk by chunk: randn x_T (unused by the loop), randn x_coarse, randn_like noise, then `steps` randn_like(known) when
    # the chunk has a known tail (every chunk but the first)
    d = dict(z_q_global=npf(z_q_global), final_mel=npf(final), params=np.array([steps, strength, scale]),
             geometry=np.array([total_frames, chunk_frames, overlap_frames, hop_len, cfg.sample_rate]))
    it = iter(rt.draws)
    for c in range(num_chunks):
        kind, _ = next(it); assert kind == "randn"
        kind, xc = next(it); assert kind == "randn"
        kind, nz = next(it); assert kind == "randn_like" and nz.shape == xc.shape
        d[f"c{c}_x_coarse"], d[f"c{c}_noise"] = npf(xc), npf(nz)
        if c > 0:
            ks = [next(it)[1] for _ in range(steps)]
            assert all(tuple(k.shape) == (1, overlap_frames, cfg.n_mels) for k in ks)
            d[f"c{c}_noise_k"] = npf(torch.stack(ks))
        d[f"c{c}_mean"], d[f"c{c}_std"] = npf(stats[c][0]), npf(stats[c][1])
    assert next(it, None) is None
    # latent slices the loop took (start_lat:end_lat) -- restated index arithmetic of :308-317, stored so that the test does not
    # depend on float rounding of the seconds arithmetic
    lat = []
    for c in range(num_chunks):
        s0, s1 = c * hop_samples, c * hop_samples + chunk_samples
        lat.append([int(s0 / cfg.sample_rate * 16000) // 320, int(s1 / cfg.sample_rate * 16000) // 320])
    d["latent_slices"] = np.array(lat)
    np.savez_compressed(os.path.join(OUT, "longform_stitch.npz"), **d)
    print("longform_stitch: final_mel", tuple(final.shape), "range", float(final.min()), float(final.max()), "latent slices", lat)


@torch.no_grad()
def bf16_sampler():
    cfg = ref.CFG(hidden=64, layers=2, heads=2, device="cpu")
    dec = make_decoder(cfg, seed=2)
    sch = ref.DiffusionSchedule(cfg.diff_steps)
    inf = ref.EdgeInference(cfg, sch, torch.nn.Identity(), dec)
    B, S = 6, 48
    sem = rnd_idx((B, S), cfg.codebook_size, 91, 0)
    x_T = rnd((B, 2 * S, 80), 91, 1, 1.0)
    real_randn = torch.randn

    def fake_randn(*a, **k):  # generate_mel draws its start noise from the global RNG (inference.py:33): inject ours
        return x_T.clone()

    outs = {}
    for tag, ctx in (("f32", None), ("bf16", torch.autocast("cpu", dtype=torch.bfloat16))):
        torch.randn = fake_randn
        try:
            if ctx is None:
                outs[tag] = inf.generate_mel(sem, 4).float()
            else:
                with ctx:
                    outs[tag] = inf.generate_mel(sem, 4).float()
        finally:
            torch.randn = real_randn
    # the t = 999 band needs eps of the first step: one fp32 decoder call
    eps0 = dec(x_T, torch.full((B,), 999), sem, torch.zeros(B, dtype=torch.long))
    err = (outs["bf16"] - outs["f32"]).abs()
    print(f"bf16_sampler: reference autocast vs fp32: median {float(err.median()):.3e} p99 {float(err.flatten().kthvalue(int(0.99 * err.numel())).values):.3e} max {float(err.max()):.3e}")
    np.savez_compressed(os.path.join(OUT, "bf16_sampler.npz"), sem_idx=npf(sem), x_T=npf(x_T), out_f32=npf(outs["f32"]),
                        out_autocast=npf(outs["bf16"]), eps0=npf(eps0), cfg=np.array([64, 2, 2]))


@torch.no_grad()
def dsconv_stride():
    """DepthwiseSeparableConv with stride > 1 (layers/conv.py:33-41), plus one case outside the fused kernel's shape class."""
    from make_golden import RefDSConv
    d = {}
    for tag, (ci, co, T_, ks, st) in {"s2": (80, 160, 64, 3, 2), "s3": (24, 40, 37, 5, 3), "wide": (96, 176, 50, 3, 2), "long": (16, 32, 700, 3, 1)}.items():
        m = RefDSConv(ci, co, kernel_size=ks, stride=st).eval()
        m.depthwise.weight.copy_(rnd(tuple(m.depthwise.weight.shape), 19, 0, 0.6))
        m.pointwise.weight.copy_(rnd(tuple(m.pointwise.weight.shape), 19, 1, 0.2))
        m.pointwise.bias.copy_(rnd(tuple(m.pointwise.bias.shape), 19, 2, 0.1))
        m.norm.weight.copy_(1.0 + rnd(tuple(m.norm.weight.shape), 19, 3, 0.2))
        m.norm.bias.copy_(rnd(tuple(m.norm.bias.shape), 19, 4, 0.1))
        xin = rnd((2, ci, T_), 19, 5, 1.0)
        d.update({f"{tag}_x": npf(xin), f"{tag}_y": npf(m(xin)), f"{tag}_dw": npf(m.depthwise.weight), f"{tag}_pw": npf(m.pointwise.weight),
                  f"{tag}_pb": npf(m.pointwise.bias), f"{tag}_gw": npf(m.norm.weight), f"{tag}_gb": npf(m.norm.bias),
                  f"{tag}_groups": np.array(m.norm.num_groups), f"{tag}_stride": np.array(st)})
    np.savez_compressed(os.path.join(OUT, "dsconv_stride.npz"), **d)


@torch.no_grad()
def forward_ffn_mult():
    """Decoder forwards with ffn_mult = 4 and 1 (config.py:99; layers/transformer.py:32-45: hidden width ffn_mult * H)."""
    d = {}
    for tag, mult in (("m4", 4), ("m1", 1)):
        cfg = ref.CFG(hidden=32, heads=2, layers=2, ffn_mult=mult, device="cpu")
        dec = make_decoder(cfg, seed=8)
        B, T, S = 2, 40, 20
        xt = rnd((B, T, 80), 63, mult, 1.5)
        tt = torch.tensor([850, 120])
        si = torch.tensor([2, 0])
        sem = rnd_idx((B, S), cfg.codebook_size, 63, 10 + mult)
        d.update({f"{tag}_x_t": npf(xt), f"{tag}_t": npf(tt), f"{tag}_step_idx": npf(si), f"{tag}_sem_idx": npf(sem),
                  f"{tag}_eps": npf(dec(xt, tt, sem, si))})
    np.savez_compressed(os.path.join(OUT, "forward_ffn_mult.npz"), **d)


FIXTURES = {"forward_ffn_mult": forward_ffn_mult, "longform_stitch": longform_stitch, "bf16_sampler": bf16_sampler, "dsconv_stride": dsconv_stride}


if __name__ == "__main__":
    names = sys.argv[1:] or list(FIXTURES)
    for n in names:
        FIXTURES[n]()
        f = os.path.join(OUT, n + ".npz")
        print(f"{n}.npz: {os.path.getsize(f) / 1024:.0f} KiB")
