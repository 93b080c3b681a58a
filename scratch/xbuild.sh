#!/bin/bash
# usage: xbuild.sh <out.so> <flags...>: experiment build of the library (default decoder fp32 instance only) + register report.
# Temporaries and the device assembly stay in /tmp/xb (asm_blocks.py reads /tmp/xb/*.s); only the .so lands in the repo.
out=$1; shift
mkdir -p /tmp/xb && cd /tmp/xb
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -mllvm -amdgpu-mfma-vgpr-form -Rpass-analysis=kernel-resource-usage -save-temps=obj -DEDTTS_EXPERIMENTS ${FAST--DEDTTS_FAST_BUILD} "$@" /root/repo/edge-diffusion-tts_amd/csrc/edtts_kernels.hip -o /tmp/xb/lib.so 2> /tmp/xb/build.log || { grep -m5 "error" -A5 /tmp/xb/build.log; exit 1; }
cp /tmp/xb/lib.so /root/repo/$out
python3 - <<'PY'
import sys
sys.path.insert(0,'/root/repo')
import __graft_entry__ as g
res=g.kernel_resources(open('/tmp/xb/build.log').read())
for k,v in res.items():
    if ("k_layer" in k or "k_prologue" in k or "dsconv_fused" in k):
        print(k[:80], v)
PY
