"""Diagnostic (-DEDTTS_EXPERIMENTS -DEDTTS_STAMPS build): s_memtime stamps of one interior wave of the fp32 layer kernel.
EDTTS_LIB=<stamps build> python scratch/stamps_f32.py   -> per-phase and per-attention-step cycle counts (BASELINE config 2 shape)"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "edge-diffusion-tts_amd"), REPO]
import torch
from edge_diffusion_tts_amd import CFG, EdgeDiffusionDecoder, synth_state_dict, native
os.chdir("/tmp")
cfg = CFG(device="cuda")
dec = EdgeDiffusionDecoder(cfg); dec.load_state_dict(synth_state_dict(cfg, 0)); dec = dec.cuda().eval()
gen = torch.Generator().manual_seed(0)
B, T, S = 256, 512, 256
x = torch.randn(B, T, 80, generator=gen).cuda(); sem = torch.randint(0, 512, (B, S), generator=gen).cuda()
t = torch.full((B,), 500).cuda(); si = torch.zeros(B, dtype=torch.long).cuda()
buf = torch.zeros(128 * 4, dtype=torch.int64, device="cuda")
L = native.lib()
L.edtts_debug_set_stamps.argtypes = [C.c_void_p]
for _ in range(3): dec(x, t, sem, si)
L.edtts_debug_set_stamps(buf.data_ptr())
names = ["load h + park + self-attn (4 heads: 5 steps + proj) + residual", "norm2 + q_proj (10 phases)", "cross-attn (4 heads x 8 steps + out_proj) + residual",
         "norm3 + FFN (20 up/down pairs) + residual", "tail (store h, norm, QKV 15 pair phases | final norm + out_proj + DDIM)", "store drain"]
mf = [4 * (5 * 88 + 200), 800, 4 * (8 * 88 + 200), 4800, None, 0]
sel = [(int(b_), int(w_), int(h_)) for b_, w_, h_ in (x.split(":") for x in os.environ.get("STAMP_SELECT", "16:0:1").split(","))]
for blk, wv, hd in sel:
    os.environ.update(EDTTS_STAMP_BLOCK=str(blk), EDTTS_STAMP_WAVE=str(wv), EDTTS_STAMP_HEAD=str(hd))
    buf.zero_()
    dec(x, t, sem, si); torch.cuda.synchronize()
    full = buf.cpu().view(4, 128)
    print(f"==== block {blk} wave {wv} head {hd}")
    for l in (int(v) for v in os.environ.get("STAMP_LAYERS", "2").split(",")):
        st = full[l, :7]
        d = (st[1:] - st[:-1]).tolist()
        tot = int(st[6] - st[0])
        print(f"layer {l}: total {tot} cycles")
        for n, v, m in zip(names, d, mf):
            m = (2400 if l < 3 else 400) if m is None else m
            print(f"   {n:75s} {v:8d}  {100.0*v/tot:5.1f} %   MFMA {m:5d} x32 = {32*m:7d}  -> {100.0*32*m/max(v,1):5.1f} % busy")
        for name, off, n in (("self-attention", 8, 5), ("cross-attention", 40, 8)):
            fs = full[l, off:off + 4 * n + 2]
            print(f"   {name}, per step: QK+K loads | softmax | PV+V loads   (ideal 40x32=1280 | - | 48x32=1536)")
            for i in range(n):
                a, b_, c, d2 = fs[4 * i:4 * i + 4].tolist()
                print(f"      step {i}: {b_-a:6d} {c-b_:6d} {d2-c:6d}   total {d2-a}")
            print(f"      normalise {int(fs[4*n]-fs[4*n-1])}, projection phases (200 MFMA = 6400) {int(fs[4*n+1]-fs[4*n])}")
