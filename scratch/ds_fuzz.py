"""Random-shape check of the one-kernel DepthwiseSeparableConv forms (C_out = 160, GroupNorm(8): the group-pipelined kernel) against the CPU oracle."""
import os, sys, random
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "edge-diffusion-tts_amd"), REPO]
import torch
from edge_diffusion_tts_amd import DepthwiseSeparableConv
from oracle import edtts_oracle as O
random.seed(int(os.environ.get("SEED", "1")))
gen = torch.Generator().manual_seed(7)
worst = 0.0
for it in range(int(os.environ.get("N", "40"))):
    B, ci, T, ks = random.randint(1, 5), random.randint(1, 80), random.randint(1, 512), random.choice([1, 3, 5])
    m = DepthwiseSeparableConv(ci, 160, kernel_size=ks)
    with torch.no_grad():
        m.norm.weight.copy_(torch.randn(160, generator=gen)); m.norm.bias.copy_(torch.randn(160, generator=gen)); m.pointwise.bias.copy_(torch.randn(160, generator=gen))
    x = torch.randn(B, ci, T, generator=gen) * random.choice([0.1, 1.0, 5.0]) + random.choice([0.0, 0.0, 3.0])
    ref = O.dsconv_forward(x.double(), m.depthwise.weight.double(), m.pointwise.weight.double(), m.pointwise.bias.double(), m.norm.weight.double(), m.norm.bias.double(), m.groups).float()
    ref32 = O.dsconv_forward(x, m.depthwise.weight, m.pointwise.weight, m.pointwise.bias, m.norm.weight, m.norm.bias, m.groups)
    y = m.to("cuda")(x.to("cuda")).cpu()
    e, e32 = float((y - ref).abs().max()), float((ref32 - ref).abs().max())
    worst = max(worst, e / max(2e-5, 3 * e32))
    flag = "" if e < max(2e-5, 3 * e32) else "   <-- FAIL"
    print(f"B={B} C_in={ci:2d} T={T:3d} k={ks}: {e:.2e} (fp32 oracle {e32:.2e}){flag}")
print("worst ratio to the bar:", round(worst, 3))
