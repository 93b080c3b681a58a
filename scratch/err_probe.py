"""Forward / end-to-end error of one library build against the reference goldens and the fp64 arbiter, plus a quick timing.
Usage (GPU box): EDTTS_LIB=<path to .so> python scratch/err_probe.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "edge-diffusion-tts_amd"), REPO, os.path.join(REPO, "tests")]
import torch
from conftest import load_golden
from edge_diffusion_tts_amd import CFG, DiffusionSchedule, EdgeDiffusionDecoder, EdgeInference, synth_state_dict, native
from oracle import edtts_oracle as O

os.chdir("/tmp")
print("lib:", native.LIB_PATH)
cfg = CFG(device="cuda")
sd = synth_state_dict(cfg, 0)
dec = EdgeDiffusionDecoder(cfg); dec.load_state_dict(sd); dec = dec.cuda().eval()
g = load_golden("forward_cfg")
e = dec(g["x_t"].cuda(), g["t"].cuda(), g["sem_idx"].cuda(), g["step_idx"].cuda()).cpu().double()
e64 = O.decoder_forward(O.cast_sd(sd, torch.float64), g["x_t"].double(), g["t"], g["sem_idx"], g["step_idx"])
def st(a, b): d = (a.double() - b.double()); return f"max {float(d.abs().max()):.2e} rms {float(d.pow(2).mean().sqrt()):.2e}"
print("forward_cfg: ours vs ref32:", st(e, g["eps"]), "| ours vs fp64:", st(e, e64), "| ref32 vs fp64:", st(g["eps"], e64))
# bigger sample for statistics
gen = torch.Generator().manual_seed(0)
B, S = 4, 128
x = torch.randn(B, 2 * S, 80, generator=gen); sem = torch.randint(0, 512, (B, S), generator=gen)
t = torch.tensor([999, 749, 499, 249]); si = torch.tensor([0, 1, 2, 3])
e = dec(x.cuda(), t.cuda(), sem.cuda(), si.cuda()).cpu()
r32 = O.decoder_forward(sd, x, t, sem, si)
r64 = O.decoder_forward(O.cast_sd(sd, torch.float64), x.double(), t, sem, si)
print("B4 T256: ours vs fp64:", st(e, r64), "| oracle32 vs fp64:", st(r32, r64))
# end to end
gg = load_golden("generate_cfg1")
sch = DiffusionSchedule(1000).to("cuda")
infer = EdgeInference(cfg, sch, torch.nn.Identity(), dec)
out = infer.generate_mel(gg["sem_idx"].cuda(), 4, x_T=gg["x_T"].cuda()).cpu()
ab = O.schedule_tables(1000)["alpha_bar"][999].double()
band = (gg["x_T"].double() - torch.sqrt(1 - ab) * gg["eps0"].double()).abs() < 3.0 * torch.sqrt(ab) * 4
err = (out.double() - gg["out"].double()).abs()
print(f"e2e cfg1 vs ref32: in-band n={int(band.sum())} max {float(err[band].max()):.2e}; out-of-band max {float(err[~band].max()):.2e} n>1e-3 {int((err[~band] > 1e-3).sum())}")
# timing at config 2
B, S = 256, 256
sem = torch.randint(0, 512, (B, S), generator=gen).cuda(); x = torch.randn(B, 2 * S, 80, generator=gen).cuda()
for _ in range(3): infer.generate_mel(sem, 4, x_T=x)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); infer.generate_mel(sem, 4, x_T=x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort(); print(f"generate_mel B=256 T=512: median {ts[5]*1e3:.3f} ms, min {ts[0]*1e3:.3f} ms -> {B*512/ts[5]/1e6:.3f} M frames/s")
