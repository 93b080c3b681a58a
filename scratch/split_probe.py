"""bf16 split layer (EDTTS16_SPLIT=1) vs the fused launch: saves outputs for a bitwise comparison, prints the call time."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "edge-diffusion-tts_amd"), REPO, os.path.join(REPO, "tests")]
import numpy as np, torch
from conftest import load_golden
from edge_diffusion_tts_amd import CFG, DiffusionSchedule, EdgeDiffusionDecoder, EdgeInference, synth_state_dict
tag = "split" if os.environ.get("EDTTS16_SPLIT") == "1" else "fused"
out_dir = os.path.join(REPO, "gpurun_out"); os.makedirs(out_dir, exist_ok=True)
gen = torch.Generator().manual_seed(0)
res = {}
cfg = CFG(hidden=64, heads=2, layers=2, attn_window_size=9, device="cuda")
sd = synth_state_dict(cfg, 7)
d16 = EdgeDiffusionDecoder(cfg, compute_dtype="bf16"); d16.load_state_dict(sd); d16 = d16.cuda().eval()
x = torch.randn(3, 154, 80, generator=gen); sem = torch.randint(0, 512, (3, 77), generator=gen)
t = torch.randint(0, 1000, (3,), generator=gen); si = torch.randint(0, 16, (3,), generator=gen)
res["tiny"] = d16(x.cuda(), t.cuda(), sem.cuda(), si.cuda()).cpu().numpy()
g = load_golden("forward_cfg3")
cfg3 = CFG(hidden=256, layers=8, heads=8, device="cuda")
sd3 = synth_state_dict(cfg3, 1, max_pos=1024)
dec = EdgeDiffusionDecoder(cfg3, max_len=1024, compute_dtype="bf16"); dec.load_state_dict(sd3); dec = dec.cuda().eval()
e = dec(g["x_t"].cuda(), g["t"].cuda(), g["sem_idx"].cuda(), g["step_idx"].cuda()).cpu()
res["cfg3"] = e.numpy()
d = (e.double() - g["eps"].double())
print(f"{tag}: cfg3 vs reference fp32: max {float(d.abs().max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e}")
sch = DiffusionSchedule(1000).to("cuda")
infer = EdgeInference(cfg3, sch, torch.nn.Identity(), dec)
B = 256
sem = torch.randint(0, 512, (B, 512), generator=gen).cuda()
for _ in range(2): infer.generate_mel(sem, 4, seed=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for i in range(n): out = infer.generate_mel(sem, 4, seed=2 + i)
torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / n
res["mel"] = out[:4].cpu().numpy()
print(f"{tag}: cfg3 bf16 B={B} T=1024: {dtm*1e3:.2f} ms/call, finite {bool(torch.isfinite(out).all())}")
np.savez(os.path.join(out_dir, f"split_probe_{tag}.npz"), **res)
other = os.path.join(out_dir, f"split_probe_{'fused' if tag == 'split' else 'split'}.npz")
if os.path.exists(other):
    o = np.load(other)
    for k in res: print(f"  {k}: bitwise equal to the other mode: {bool(np.array_equal(res[k], o[k]))}  max diff {float(np.abs(res[k] - o[k]).max()):.3e}")
