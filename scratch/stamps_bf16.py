"""Diagnostic (-DEDTTS_STAMPS build): cycle stamps of block 0 / wave 0 of the bf16 layer kernel, per layer.
EDTTS_LIB=<stamps build> python scratch/stamps_bf16.py"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "edge-diffusion-tts_amd"), REPO]
import torch
from edge_diffusion_tts_amd import CFG, EdgeDiffusionDecoder, synth_state_dict, native
os.chdir("/tmp")
cfg = CFG(hidden=256, layers=8, heads=8, device="cuda")
dec = EdgeDiffusionDecoder(cfg, max_len=1024, compute_dtype="bf16"); dec.load_state_dict(synth_state_dict(cfg, 1, max_pos=1024)); dec = dec.cuda().eval()
gen = torch.Generator().manual_seed(0)
B, T, S = 256, 1024, 512
x = torch.randn(B, T, 80, generator=gen).cuda(); sem = torch.randint(0, 512, (B, S), generator=gen).cuda()
t = torch.full((B,), 500).cuda(); si = torch.zeros(B, dtype=torch.long).cuda()
buf = torch.zeros(128 * 8, dtype=torch.int64, device="cuda")
L = native.lib()
L.edtts_debug_set_stamps.argtypes = [C.c_void_p]
for _ in range(2): dec(x, t, sem, si)
L.edtts_debug_set_stamps(buf.data_ptr())
dec(x, t, sem, si); torch.cuda.synchronize()
full = buf.cpu().view(8, 128)
st = full[:, :7]
names = ["self-attn (8 heads: steps + proj phase)", "norm2 + q_proj (8 phases)", "cross-attn (8 heads x 16 steps + out_proj phase)", "norm3 + FFN (48 phases)", "tail: store h, norm, QKV (24 phases)", "drain"]
for l in (1, 4):
    d = (st[l, 1:] - st[l, :-1]).tolist()
    tot = int(st[l, 6] - st[l, 0])
    print(f"layer {l}: total {tot} memtime ticks (100 MHz constant clock? see below)")
    for n, v in zip(names, d): print(f"   {n:55s} {v:9d}  {100.0*v/tot:5.1f} %")

t = full[4, 120:124].tolist() + [int(st[4, 5])]
print(f"layer 4 tail: entry->q {t[1]-int(st[4,4])} (store h, norm) | q phases {t[2]-t[1]} | k phases {t[3]-t[2]} | v^T phases {t[4]-t[3]}")
# fine-grained: first head pair of layer 4, 4 stamps per step: [start | after QK issue + K request | after softmax + pack | after PV issue + V request]
for name, off, n in (("self-attention", 8, 5), ("cross-attention", 40, 16)):
    fs = full[4, off:off + 4 * n].view(n, 4)
    print(f"{name}, heads 0+1, per step: QK+loads | softmax | PV+loads | (gap to next step start)")
    for i in range(n):
        a, b, c, d = fs[i].tolist()
        nxt = fs[i + 1, 0].item() if i < n - 1 else d
        print(f"  step {i:2d}: {b-a:6d} {c-b:6d} {d-c:6d} {nxt-d:6d}   total {nxt-a}")
