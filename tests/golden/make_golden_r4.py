#!/usr/bin/env python3
"""Round-4 addition to tests/golden/*.npz, made by running the REFERENCE on CPU (build container only).

    python tests/golden/make_golden_r4.py

bf16_sampler_cfg3   the reference's own generate_mel under torch.autocast("cpu", bfloat16) next to its fp32 run at BASELINE config 3's
                    decoder shape (hidden 256, 8 layers, 8 heads of 32; one utterance of T = 512 frames, 4-step DDIM): the error
                    distribution a bf16 implementation of THIS decoder has by the reference's own standard.  Inputs + outputs only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402,F401  (sets sys.path for the reference + this repo's synth module, chdirs to a scratch dir)
from make_golden import OUT, make_decoder, npf, ref, rnd, rnd_idx  # noqa: E402


@torch.no_grad()
def bf16_sampler_cfg3():
    cfg = ref.CFG(hidden=256, layers=8, heads=8, device="cpu")
    dec = make_decoder(cfg, seed=1)
    sch = ref.DiffusionSchedule(cfg.diff_steps)
    inf = ref.EdgeInference(cfg, sch, torch.nn.Identity(), dec)
    B, S = 1, 256
    sem = rnd_idx((B, S), cfg.codebook_size, 93, 0)
    x_T = rnd((B, 2 * S, 80), 93, 1, 1.0)
    real_randn = torch.randn

    def fake_randn(*a, **k):  # generate_mel draws its start noise from the global RNG (inference.py:33): inject ours
        return x_T.clone()

    outs = {}
    for tag, ctx in (("f32", None), ("bf16", torch.autocast("cpu", dtype=torch.bfloat16))):
        torch.randn = fake_randn
        try:
            if ctx is None:
                outs[tag] = inf.generate_mel(sem, 4).float()
            else:
                with ctx:
                    outs[tag] = inf.generate_mel(sem, 4).float()
        finally:
            torch.randn = real_randn
    eps0 = dec(x_T, torch.full((B,), 999), sem, torch.zeros(B, dtype=torch.long))  # the t = 999 band needs eps of the first step
    err = (outs["bf16"] - outs["f32"]).abs()
    q = lambda p: float(err.flatten().kthvalue(max(1, int(p * err.numel()))).values)
    print(f"bf16_sampler_cfg3: reference autocast vs fp32: median {q(0.5):.3e} p90 {q(0.9):.3e} p99 {q(0.99):.3e} max {float(err.max()):.3e}")
    np.savez_compressed(os.path.join(OUT, "bf16_sampler_cfg3.npz"), sem_idx=npf(sem), x_T=npf(x_T), out_f32=npf(outs["f32"]),
                        out_autocast=npf(outs["bf16"]), eps0=npf(eps0), cfg=np.array([256, 8, 8]))


if __name__ == "__main__":
    bf16_sampler_cfg3()
    f = os.path.join(OUT, "bf16_sampler_cfg3.npz")
    print(f"bf16_sampler_cfg3.npz: {os.path.getsize(f) / 1024:.0f} KiB")
