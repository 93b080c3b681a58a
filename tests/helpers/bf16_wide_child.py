"""Child process of test_bf16_wide_instance (tests/test_gpu_parity.py): the library reads EDTTS16_WIDE once per process, so the
64-frame bf16 instance is exercised in a process of its own.  Writes its outputs to the .npz named on the command line."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "tests"), REPO]
from conftest import load_golden  # noqa: E402
from edge_diffusion_tts_amd import (CFG, DiffusionSchedule, EdgeDiffusionDecoder, EdgeInference,  # noqa: E402
                                    synth_state_dict)

assert os.environ.get("EDTTS16_WIDE") == "1"
out = sys.argv[1]
DEV = "cuda"
cfg = CFG(hidden=256, layers=8, heads=8, device=DEV)
dec = EdgeDiffusionDecoder(cfg, max_len=1024, compute_dtype="bf16")
dec.load_state_dict(synth_state_dict(cfg, 1, max_pos=1024))
dec = dec.to(DEV).eval()
g = load_golden("forward_cfg3")
eps = dec(g["x_t"].to(DEV), g["t"].to(DEV), g["sem_idx"].to(DEV), g["step_idx"].to(DEV)).cpu()
# the sampler on a batch whose utterances straddle blocks (T = 768: three 256-frame blocks per utterance), each probed utterance
# also run alone: a block's QKV tail writes the next layer's images while its neighbours still read this layer's
infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), torch.nn.Identity(), dec)
gen = torch.Generator().manual_seed(22)
B, S = 24, 384
sem = torch.randint(0, 512, (B, S), generator=gen).to(DEV)
x = torch.randn(B, 2 * S, 80, generator=gen).to(DEV)
big = infer.generate_mel(sem, 4, x_T=x)
again = infer.generate_mel(sem, 4, x_T=x)
alone = torch.cat([infer.generate_mel(sem[i:i + 1].contiguous(), 4, x_T=x[i:i + 1].contiguous()) for i in (0, 11, 23)])
# ragged geometries against the CPU oracle (two layers of the same width): padded frame tiles (T = 250 -> 256), a partial last
# key chunk, windows below / on / above the 64-frame group, full attention, a context longer than the utterance
from oracle import edtts_oracle as O  # noqa: E402

geo = []
for B2, T2, S2, window in [(2, 250, 100, 16), (1, 128, 33, None), (2, 320, 64, 128), (1, 64, 512, 64), (3, 192, 77, 40)]:
    cfg2 = CFG(hidden=256, heads=8, layers=2, attn_window_size=window, device=DEV)
    sd = synth_state_dict(cfg2, 5)
    d2 = EdgeDiffusionDecoder(cfg2, compute_dtype="bf16")
    d2.load_state_dict(sd)
    d2 = d2.to(DEV).eval()
    g2 = torch.Generator().manual_seed(1000 + T2)
    x2 = torch.randn(B2, T2, 80, generator=g2)
    sem2 = torch.randint(0, 512, (B2, S2), generator=g2)
    t2 = torch.randint(0, 1000, (B2,), generator=g2)
    si2 = torch.randint(0, 16, (B2,), generator=g2)
    e2 = d2(x2.to(DEV), t2.to(DEV), sem2.to(DEV), si2.to(DEV)).cpu()
    ref2 = O.decoder_forward(sd, x2, t2, sem2, si2, heads=8, window=window)
    dd = (e2.double() - ref2.double())
    geo.append([T2, S2, -1 if window is None else window, float(dd.pow(2).mean().sqrt()), float(dd.abs().max()), float(torch.isfinite(e2).all())])
np.savez(out, eps=eps.numpy(), big=big.cpu().numpy(), again=again.cpu().numpy(), alone=alone.cpu().numpy(), geo=np.array(geo))
