"""Phase timeline of k_dsconv_fused (sixteen-wave form) from a -DEDTTS_DS_STAMPS build: s_memtime of wave 0 of every block.
usage: EDTTS_LIB=scratch/lib_dsst.so python scratch/ds_stamps.py"""
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
conv(xc)
torch.cuda.synchronize()
L = native.lib()
buf = np.zeros(1024 * 16, dtype=np.uint64)
rc = L.edtts_debug_read_ds_stamps(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
st = buf.reshape(1024, 16)[:256].astype(np.int64)
names = ["start", "p0 parked", "p0 sync", "p0 taps", None, "p1 parked", "p1 sync", "p1 taps", None, "mfma done", "stats 1", "stats 2", "stores issued", "stores acked"]
rel = st - st[:, :1]  # per block, relative to its own start (the counters of different XCDs are not synchronised)
print("s_memtime ticks (shader clock, ~2.38 GHz under load) of wave 0 of every block, relative to the block's own start: median / min / max over blocks, and the phase length")
prev = None
for i, n in enumerate(names):
    if n is None: continue
    col = rel[:, i] / 2380.0
    med = np.median(col)
    print(f"{n:>14}: median {med:7.2f} us  min {col.min():7.2f}  max {col.max():7.2f}" + (f"   (+{med - prev:6.2f})" if prev is not None else ""))
    prev = med
rt0, rt1 = st[:, 14], st[:, 15]  # s_memrealtime (100 MHz, chip-wide) at the start and end of wave 0 of every block
base = rt0.min()
print(f"realtime: block starts {np.percentile(rt0 - base, [0, 50, 90, 100]) * 0.01} us (min / median / p90 / max), block ends {np.percentile(rt1 - base, [0, 50, 90, 100]) * 0.01} us")
print(f"          block lifetime median {np.median(rt1 - rt0) * 0.01:.2f} us, first start -> last end {(rt1.max() - base) * 0.01:.2f} us")
