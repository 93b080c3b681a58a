"""One-off extended fuzz of the mask / chunk-order / tile-clamping logic against the oracle (fp32 default decoder and the bf16
hidden-64 instance): many more (B, T, S, window) cases than tests/test_gpu_parity.py::test_forward_random_geometries runs."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "edge-diffusion-tts_amd"), REPO]
import numpy as np, torch
from edge_diffusion_tts_amd import CFG, EdgeDiffusionDecoder, synth_state_dict
from oracle import edtts_oracle as O
os.chdir("/tmp")
n32, n16 = int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "7")))
def stats(a, b):
    d = a.double() - b.double()
    return float(d.abs().max()), float(d.pow(2).mean().sqrt())
worst32 = worst16 = 0.0
t0 = time.time()
for case in range(n32 + n16):
    bf16 = case >= n32
    B = int(rng.integers(1, 10))
    S = int(rng.integers(1, 330 if not bf16 else 200))
    T = 2 * S
    window = [None, int(rng.integers(1, 140)), int(rng.integers(1, 20)), 64, 16, 32, 48][int(rng.integers(0, 7))]
    kw = dict(hidden=64, heads=2, layers=2) if bf16 else {}
    cfg = CFG(device="cuda", attn_window_size=window, **kw)
    sd = synth_state_dict(cfg, int(rng.integers(0, 5)))
    dec = EdgeDiffusionDecoder(cfg, compute_dtype="bf16" if bf16 else "f32"); dec.load_state_dict(sd); dec = dec.cuda().eval()
    gen = torch.Generator().manual_seed(1000 + case)
    x = torch.randn(B, T, 80, generator=gen); sem = torch.randint(0, 512, (B, S), generator=gen)
    t = torch.randint(0, 1000, (B,), generator=gen); si = torch.randint(0, 16, (B,), generator=gen)
    e = dec(x.cuda(), t.cuda(), sem.cuda(), si.cuda()).cpu()
    ref = O.decoder_forward(sd, x, t, sem, si, heads=cfg.heads, window=window)
    mx, rm = stats(e, ref)
    ok = bool(torch.isfinite(e).all()) and ((rm < 2.6e-3 and mx < 2e-2) if bf16 else mx < 2e-5)
    if bf16: worst16 = max(worst16, rm)
    else: worst32 = max(worst32, mx)
    if not ok or case % 20 == 0:
        print(f"case {case} {'bf16' if bf16 else 'f32'} B={B} T={T} S={S} window={window}: max {mx:.2e} rms {rm:.2e} {'ok' if ok else 'FAIL'} ({time.time()-t0:.0f} s)", flush=True)
    if not ok: sys.exit(1)
print(f"fuzz ok: {n32} fp32 cases worst max-abs {worst32:.2e}; {n16} bf16 cases worst rms {worst16:.2e}")
