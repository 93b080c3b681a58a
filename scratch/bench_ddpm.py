"""BASELINE config 5: 1000-step DDPM teacher sampler, B=64, T=512, one GPU -- eager launches vs one captured hipGraph."""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "edge-diffusion-tts_amd")); sys.path.insert(0, REPO)
import torch
from edge_diffusion_tts_amd import CFG, DiffusionSchedule, EdgeDiffusionDecoder, EdgeInference, synth_state_dict
cfg = CFG(device="cuda"); dec = EdgeDiffusionDecoder(cfg); dec.load_state_dict(synth_state_dict(cfg, 0)); dec = dec.cuda().eval()
infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to("cuda"), None, dec)
B, S = 64, 256
g = torch.Generator().manual_seed(0)
sem = torch.randint(0, 512, (B, S), generator=g).cuda(); x = torch.randn(B, 2 * S, 80, generator=g).cuda()
steps = int(os.environ.get("STEPS", "1000"))
out = infer.sample_ddpm(sem, steps, x_T=x, seed=1); torch.cuda.synchronize()
t0 = time.perf_counter(); out = infer.sample_ddpm(sem, steps, x_T=x, seed=1); torch.cuda.synchronize(); eager = time.perf_counter() - t0
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out_g = infer.sample_ddpm(sem, steps, x_T=x, seed=1)
gr.replay(); torch.cuda.synchronize()
t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize(); graph = time.perf_counter() - t0
flop = steps * B * 2 * S * 3473600.0
print(json.dumps({"config": f"DDPM {steps} steps B={B} T={2*S}", "eager_s": eager, "graph_s": graph, "equal": bool(torch.equal(out, out_g)),
                  "finite": bool(torch.isfinite(out).all()), "frames_per_s_graph": B * 2 * S / graph, "step_ms_graph": graph / steps * 1e3,
                  "algorithmic_tflops_graph": flop / graph / 1e12}))
