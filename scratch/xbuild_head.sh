#!/bin/bash
# usage: xbuild_head.sh <out.so> [flags]: the same experiment build as xbuild.sh, but from the sources of git HEAD (the A/B base)
out=$1; shift
rm -rf /tmp/headsrc && mkdir -p /tmp/headsrc && git archive HEAD edge-diffusion-tts_amd/csrc include | tar -x -C /tmp/headsrc
mkdir -p /tmp/xbh && cd /tmp/xbh
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -mllvm -amdgpu-mfma-vgpr-form -DEDTTS_EXPERIMENTS ${FAST--DEDTTS_FAST_BUILD} "$@" /tmp/headsrc/edge-diffusion-tts_amd/csrc/edtts_kernels.hip -o /root/repo/$out || exit 1
echo built $out from $(cd /root/repo && git rev-parse --short HEAD)
