"""Diagnostic (-DEDTTS_EXPERIMENTS -DEDTTS_WAVELOG build): start / end of EVERY wave of the layer launches of one generate_mel step
(s_memrealtime = 100 MHz wall clock, s_memtime = shader cycles) -> slot occupancy over time, in-kernel clock, tail shape.
EDTTS_LIB=<wavelog build> python scratch/wavelog.py [substreams]"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "edge-diffusion-tts_amd"), REPO]
import numpy as np, torch
from edge_diffusion_tts_amd import CFG, DiffusionSchedule, EdgeDiffusionDecoder, EdgeInference, synth_state_dict, native
os.chdir("/tmp")
subs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = CFG(device="cuda")
dec = EdgeDiffusionDecoder(cfg); dec.load_state_dict(synth_state_dict(cfg, 0)); dec = dec.cuda().eval()
infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to("cuda"), torch.nn.Identity(), dec)
B, S = 256, 256
sem = torch.randint(0, 512, (B, S), generator=torch.Generator().manual_seed(0)).cuda()
native.set_substreams(subs)
L = native.lib()
L.edtts_debug_set_wavelog.argtypes = [C.c_void_p]
for _ in range(30): infer.generate_mel(sem, 4, seed=1)   # warm (clocks settle)
torch.cuda.synchronize()
buf = torch.zeros(4 * 8192 * 6, dtype=torch.int64, device="cuda")
L.edtts_debug_set_wavelog(buf.data_ptr())
infer.generate_mel(sem, 4, seed=2); torch.cuda.synchronize()   # the log keeps the LAST step's four layer launches
L.edtts_debug_set_wavelog(None)
log = buf.cpu().numpy().reshape(4, 8192, 6)
for l in range(4):
    w = log[l][log[l][:, 2] > 0]
    r0, c0, r1, c1 = (w[:, i].astype(np.float64) for i in range(4))
    t0 = r0.min()
    start, end = (r0 - t0) / 100.0, (r1 - t0) / 100.0   # microseconds
    life_us, life_cyc = end - start, c1 - c0
    clk = life_cyc / life_us / 1e3
    span = end.max()
    simd = (w[:, 4] >> 4) & 3; cu = (w[:, 4] >> 8) & 15; se = (w[:, 4] >> 13) & 7; sh = (w[:, 4] >> 12) & 1; xcc = w[:, 5] & 15
    slot = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    slot = slot * 4 + simd
    nslots = len(np.unique(slot))
    print(f"layer {l}: {len(w)} waves on {nslots} SIMDs, span {span:.1f} us; wave lifetime {life_us.mean():.1f} us (min {life_us.min():.1f}, max {life_us.max():.1f}) = "
          f"{life_cyc.mean():.0f} cycles; in-kernel clock {clk.mean():.3f} GHz; sum(lifetime)/span = {life_us.sum()/span:.0f} waves in flight on average")
    edges = np.linspace(0, span, 21)
    occ = [((start < b) & (end > a)).sum() and (np.minimum(end, b) - np.maximum(start, a)).clip(0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
    print("   waves in flight per 5 % of the span: " + " ".join(f"{o:.0f}" for o in occ))
    per = np.bincount(np.unique(slot, return_inverse=True)[1])
    print(f"   waves per SIMD: min {per.min()} max {per.max()}; last start {start.max():.1f} us; first end {end.min():.1f} us; ends after 95% of span: {(end > 0.95*span).sum()}")
