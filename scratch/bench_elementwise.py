"""HBM-roofline evidence for the standalone elementwise / conv kernels: GB/s at [256, 512, 80]."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "edge-diffusion-tts_amd"), REPO]
import torch
from edge_diffusion_tts_amd import DepthwiseSeparableConv, DiffusionSchedule, native
os.chdir("/tmp")
dev = "cuda"
B, T, M = 1024, 512, 80  # 168 MB per tensor: past the 256 MiB Infinity Cache
g = torch.Generator().manual_seed(0)
x, eps, nz = (torch.randn(B, T, M, generator=g).to(dev) for _ in range(3))
t = torch.randint(1, 1000, (B,), generator=g).to(dev); tp = (t - 250).clamp(min=0)
sch = DiffusionSchedule(1000).to(dev)
def timeit(fn, n=50):
    for _ in range(5): fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ts[n // 2] * 1e-3
res = {}
nb = x.numel() * 4
s = timeit(lambda: sch.get_ddim_step(x, t, tp, eps)); res["k_ddim (read x, eps; write x_prev, x0)"] = (4 * nb, s)
s = timeit(lambda: sch.get_ddim_step(x, t, tp, eps, eta=0.5, noise=nz)); res["k_ddim eta>0 (+ noise read)"] = (5 * nb, s)
s = timeit(lambda: sch.ddpm_step(x, t, eps, noise=nz)); res["k_ddpm (read x, eps, noise; write x_prev)"] = (4 * nb, s)
s = timeit(lambda: native.randn((B, T, M), dev, seed=1)); res["k_randn (Philox + Box-Muller; write)"] = (nb, s)
s = timeit(lambda: x + eps); res["torch add (reference point: read 2, write 1)"] = (3 * nb, s)
# depthwise-separable conv, the north star's shape: 80 -> 160 channels on [256, 80, 512]
conv = DepthwiseSeparableConv(80, 160, 3).to(dev)
B = 256
xc = torch.randn(B, 80, T, generator=g).to(dev)
s_eager = timeit(lambda: conv(xc), 20)  # event to event around eager calls: kernel + launch gap + the Python wrapper
# the kernel without the host in the way: 20 calls captured in one hipGraph, replayed (rocprofv3's kernel duration agrees to 0.2 us)
gr, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(3): conv(xc)
    torch.cuda.synchronize()
    with torch.cuda.graph(gr, stream=st):
        for _ in range(20): yc = conv(xc)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = []
for _ in range(11):
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    reps.append(e0.elapsed_time(e1) / 20)
s = sorted(reps)[5] * 1e-3
alg = (B * 80 * T + B * 160 * T) * 4  # read x once, write y once
res["dsconv 80->160 k=3 (algorithmic: read x, write y)"] = (alg, s)
fl = 2 * B * T * (80 * 3 + 80 * 160)
out = {k: {"bytes": b, "ms": s * 1e3, "GBps": b / s / 1e9, "frac_of_8TBps": b / s / 8e12} for k, (b, s) in res.items()}
out["dsconv 80->160 k=3 (algorithmic: read x, write y)"]["ms_eager_event_to_event"] = s_eager * 1e3
out["dsconv 80->160 k=3 (algorithmic: read x, write y)"]["TFLOPs"] = fl / res["dsconv 80->160 k=3 (algorithmic: read x, write y)"][1] / 1e12
print(json.dumps(out, indent=1))
