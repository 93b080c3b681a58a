"""Where the bf16 layer kernel spends its time: decoder forward at config-3 width with the attention work dialled down at run time."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "edge-diffusion-tts_amd"), REPO]
import torch
from edge_diffusion_tts_amd import CFG, EdgeDiffusionDecoder, synth_state_dict
os.chdir("/tmp")
gen = torch.Generator().manual_seed(0)
B, T = 256, 1024
def run(S, window, dtype="bf16"):
    cfg = CFG(hidden=256, layers=8, heads=8, attn_window_size=window, device="cuda")
    dec = EdgeDiffusionDecoder(cfg, max_len=1024, compute_dtype=dtype); dec.load_state_dict(synth_state_dict(cfg, 1, max_pos=1024)); dec = dec.cuda().eval()
    x = torch.randn(B, T, 80, generator=gen).cuda(); sem = torch.randint(0, 512, (B, S), generator=gen).cuda()
    t = torch.full((B,), 500).cuda(); si = torch.zeros(B, dtype=torch.long).cuda()
    for _ in range(2): dec(x, t, sem, si)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): dec(x, t, sem, si)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3
base = run(512, 64)
print(f"forward (8 layers + ctx + prologue), S=512 window=64: {base:.2f} ms")
a = run(32, 64); print(f"S=32 (cross-attention 1 chunk instead of 16): {a:.2f} ms -> 15 cross chunks/head cost {(base - a):.2f} ms = {(base-a)/8/15/8*1e3:.1f} us per chunk-step-launch... per layer {(base-a)/8:.3f} ms")
b = run(512, 0); print(f"window=0 (self-attention 1 chunk instead of 5): {b:.2f} ms -> 4 self chunks/head cost {(base - b):.2f} ms, per layer {(base-b)/8:.3f} ms")
c = run(32, 0); print(f"S=32, window=0 (GEMM phases + 2 chunk-steps per head): {c:.2f} ms, per layer {c/8:.3f} ms")
