#!/bin/bash
# usage: scratch/ds_ablate.sh "<flags A>" "<flags B>" ... -- one library per flag set, times the fused DepthwiseSeparableConv layer [256, 80 -> 160, 512]
set -e
SRC=edge-diffusion-tts_amd/csrc/edtts_kernels.hip
i=0
for f in "$@"; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -mllvm -amdgpu-mfma-vgpr-form -DEDTTS_EXPERIMENTS $f $SRC -o /tmp/lib_ds_$i.so & i=$((i+1)); done; wait
cat > /tmp/ds_time.py <<'PY'
import os, sys
REPO = os.environ["GRAFT_REPO_ROOT"] if "GRAFT_REPO_ROOT" in os.environ else os.getcwd()
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
print(f"{ts[n // 2] * 1e3:.1f} us")
PY
for rep in 1 2; do i=0; for f in "$@"; do
  echo -n "[$f] "; EDTTS_LIB=/tmp/lib_ds_$i.so python /tmp/ds_time.py 2>/dev/null
  i=$((i+1)); done; done
