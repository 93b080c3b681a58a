"""k_prologue / whole-call timing of the fp32 headline for a library build (EDTTS_LIB): rocprof-free, via torch events around calls."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "edge-diffusion-tts_amd"), REPO]
import torch
from edge_diffusion_tts_amd import CFG, DiffusionSchedule, EdgeDiffusionDecoder, EdgeInference, synth_state_dict, native
os.chdir("/tmp")
cfg = CFG(device="cuda")
dec = EdgeDiffusionDecoder(cfg); dec.load_state_dict(synth_state_dict(cfg, 0)); dec = dec.cuda().eval()
infer = EdgeInference(cfg, DiffusionSchedule(1000).to("cuda"), torch.nn.Identity(), dec)
gen = torch.Generator().manual_seed(0)
B, S = 256, 256
sem = torch.randint(0, 512, (B, S), generator=gen).cuda(); x = torch.randn(B, 2 * S, 80, generator=gen).cuda()
for _ in range(5): out = infer.generate_mel(sem, 4, x_T=x)
torch.cuda.synchronize()
n = 40
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    out = infer.generate_mel(sem, 4, x_T=x); ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
print(f"{os.path.basename(native.LIB_PATH)}: call median {ts[n//2]:.3f} ms min {ts[0]:.3f}; checksum {float(out.double().sum()):.6f}")
