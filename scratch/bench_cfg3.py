"""BASELINE config 3 SHAPE (hidden=256, layers=8, heads=8, B=256, T=1024, 4-step DDIM, one GPU) on the fp32 instance.
BASELINE.json names bf16 for this config; no bf16 instance exists yet, so this is the fp32 number for the same shape."""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "edge-diffusion-tts_amd")); sys.path.insert(0, REPO)
import torch
from edge_diffusion_tts_amd import CFG, DiffusionSchedule, EdgeDiffusionDecoder, EdgeInference, synth_state_dict, native
cfg = CFG(hidden=256, layers=8, heads=8, device="cuda")
dec = EdgeDiffusionDecoder(cfg, max_len=1024, max_context_len=512)
dec.load_state_dict(synth_state_dict(cfg, 0, max_pos=1024, max_ctx_pos=512)); dec = dec.cuda().eval()
infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to("cuda"), None, dec)
B, S = int(os.environ.get("B", "256")), 512
g = torch.Generator().manual_seed(0)
sem = torch.randint(0, 512, (B, S), generator=g).cuda(); x = torch.randn(B, 2 * S, 80, generator=g).cuda()
for _ in range(2): out = infer.generate_mel(sem, 4, x_T=x)
torch.cuda.synchronize()
native.profile_enable(True)
n = 5
t0 = time.perf_counter()
for _ in range(n): out = infer.generate_mel(sem, 4, x_T=x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
(ms, launches), _ = native.profile_collect()
H, L, T = 256, 8, 2 * S
nk = sum(min(T - 1, i + 64) - max(0, i - 64) + 1 for i in range(T)) / T
per_frame_layer = 18 * H * H + 4 * nk * H + 4 * S * H + 6 * H * H  # DESIGN.md 4.2, QKV tail included
flop_call = 4 * B * T * (L * per_frame_layer - 6 * H * H + 2 * 80 * H + 2 * 80 * H + 6 * H * H)
print(json.dumps({"config": f"config-3 shape H=256 L=8 heads=8 B={B} T={T}, 4-step DDIM, fp32 instance", "ms_per_call": dt * 1e3,
                  "mel_frames_per_s": B * T / dt, "finite": bool(torch.isfinite(out).all()),
                  "algorithmic_tflops": flop_call / dt / 1e12, "k_layer_avg_ms": ms / max(launches, 1), "k_layer_launches": launches}))
