"""Times the fused DepthwiseSeparableConv layer [256, 80 -> 160, 512] (HIP events around each call; run under rocprofv3 --kernel-trace
--stats for the kernel's own duration)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "edge-diffusion-tts_amd"), REPO]
import torch
from edge_diffusion_tts_amd import DepthwiseSeparableConv
g = torch.Generator().manual_seed(0)
conv = DepthwiseSeparableConv(80, 160, 3).to("cuda")
xc = torch.randn(256, 80, 512, generator=g).to("cuda")
for _ in range(5): conv(xc)
n = 40
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    conv(xc); ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
print(f"{ts[n // 2] * 1e3:.1f} us per call (median of {n})")
# the same kernel without the host in the way: 20 calls captured in a hipGraph, replayed
y = conv(xc)
gr = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): conv(xc)
    torch.cuda.synchronize()
    with torch.cuda.graph(gr, stream=s):
        for _ in range(20): y = conv(xc)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(10):
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20)
ts.sort()
print(f"{ts[5] * 1e3:.1f} us per call inside a 20-call hipGraph (median of 10 replays)")
