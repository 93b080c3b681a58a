"""Timeline of k_dsconv_grouped from a -DEDTTS_DS_STAMPS build (wave 0 of every block).  usage: EDTTS_LIB=scratch/lib_dsst.so python scratch/ds_stamps_grp.py"""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "edge-diffusion-tts_amd"), REPO]
import torch
from edge_diffusion_tts_amd import DepthwiseSeparableConv, native
g = torch.Generator().manual_seed(0)
conv = DepthwiseSeparableConv(80, 160, 3).to("cuda")
xc = torch.randn(256, 80, 512, generator=g).to("cuda")
for _ in range(5): conv(xc)
torch.cuda.synchronize()
L = native.lib()
buf = np.zeros(1024 * 16, dtype=np.uint64)
assert L.edtts_debug_read_ds_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.reshape(1024, 16)[:256].astype(np.int64)
rt0, rt1 = st[:, 14], st[:, 15]
life = np.median(rt1 - rt0) * 0.01
ticks = np.median(st[:, 13] - st[:, 0])
print(f"block lifetime (s_memrealtime) median {life:.2f} us; shader clock {ticks / life:.0f} MHz; first start -> last end {(rt1.max() - rt0.min()) * 0.01:.2f} us; start spread {(rt0.max() - rt0.min()) * 0.01:.2f} us")
SUB = os.environ.get("DS_SUB", "0") == "1"  # a -DEDTTS_DS_STAMPS=2 build: sub-phases of the step of tile 3
names = ["start", "depthwise done", "t3: step start (tile 3 MFMAs issued earlier)", "t3: sum + wave reduce", "t3: squares + wave reduce + LDS write", "t3: tile 4 MFMAs issued", "t3: barrier passed", "t3: combine done", "t3: tile stored (erf, 8 values)", None, None, None, None, "stores acked"] if SUB else ["start", "depthwise done"] + [f"tile {t} step" for t in range(10)] + ["(pass 0 parked)", "stores acked"]
rel = (st - st[:, :1]) / (ticks / life)
prev = 0.0
for i, n in enumerate(names):
    if n is None: continue
    med = float(np.median(rel[:, i]))
    print(f"{n:>48}: {med:7.2f} us (+{med - prev:5.2f})")
    prev = med
