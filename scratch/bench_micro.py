"""Compute-only price of the opt-in all-gather overlap: the local batch (256 x 512) sampled in 1 / 2 / 4 slices on one GPU."""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "edge-diffusion-tts_amd")); sys.path.insert(0, REPO)
import torch
from edge_diffusion_tts_amd import CFG, DiffusionSchedule, EdgeDiffusionDecoder, EdgeInference, synth_state_dict
cfg = CFG(device="cuda"); dec = EdgeDiffusionDecoder(cfg); dec.load_state_dict(synth_state_dict(cfg, 0)); dec = dec.cuda().eval()
infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to("cuda"), None, dec)
B, S = 256, 256
g = torch.Generator().manual_seed(0)
sem = torch.randint(0, 512, (B, S), generator=g).cuda(); x = torch.randn(B, 2 * S, 80, generator=g).cuda()
res = {}
for m in (1, 2, 4):
    Bm = B // m
    def run():
        return [infer.generate_mel(sem[i * Bm:(i + 1) * Bm], 4, x_T=x[i * Bm:(i + 1) * Bm]) for i in range(m)]
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); res[f"slices_{m}_ms"] = (time.perf_counter() - t0) / 10 * 1e3
print(json.dumps(res))
