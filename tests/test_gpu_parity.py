"""GPU parity tests: the product path (Python API -> ctypes -> libedtts_hip.so -> gfx950 kernels) against the
reference's golden vectors and the CPU oracle.  Run on the GPU box: python -m pytest tests -m gpu."""
import numpy as np
import pytest
import torch

from conftest import max_abs
from edge_diffusion_tts_amd import (CFG, DepthwiseSeparableConv, DiffusionSchedule, EdgeDiffusionDecoder, EdgeInference,
                                    synth_state_dict)
from oracle import edtts_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"

# tolerances: single forward vs the fp32 reference 1e-4 max-abs (outputs are O(1));
# DDIM / DDPM updates bit-exact; end-to-end 1e-3 outside the t=999 amplification band.
FWD_TOL = 1e-4
E2E_TOL = 1e-3


def make_decoder(cfg, seed, **kw):
    dec = EdgeDiffusionDecoder(cfg, **kw)
    dec.load_state_dict(synth_state_dict(cfg, seed, max_pos=dec.max_len, max_ctx_pos=dec.max_context_len))
    return dec.to(DEV).eval()


def cu(t):
    return None if t is None else t.to(DEV)


def test_library_is_loaded():
    import os
    import __graft_entry__ as G
    from edge_diffusion_tts_amd import native
    assert native.lib().edtts_version() >= 200
    assert torch.cuda.is_available()
    # the library under test is the one build() makes from the sources in this tree (a failed build would otherwise leave the
    # previous library in place and the suite would pass on it); EDTTS_LIB names an experiment build on purpose
    if not os.environ.get("EDTTS_LIB"):
        assert not G._stale(), "edge-diffusion-tts_amd/lib/libedtts_hip.so is older than csrc/: run __graft_entry__.build()"


def test_ddim_ddpm_bit_exact(golden):
    g = golden("steps")
    sch = DiffusionSchedule(1000).to(DEV)
    xp, x0 = sch.get_ddim_step(cu(g["x"]), cu(g["t"]), cu(g["t_prev"]), cu(g["eps"]), eta=0.0)
    assert torch.equal(xp.cpu(), g["ddim_x_prev"]) and torch.equal(x0.cpu(), g["ddim_x0"])
    xp, x0 = sch.get_ddim_step(cu(g["x"]), cu(g["t"]), cu(g["t_prev"].clamp(min=0)), cu(g["eps"]), eta=0.5, noise=cu(g["noise"]))
    assert torch.equal(xp.cpu(), g["ddim_eta_x_prev"]) and torch.equal(x0.cpu(), g["ddim_eta_x0"])
    xd = sch.ddpm_step(cu(g["x"]), cu(g["t"]), cu(g["eps"]), noise=cu(g["ddpm_noise"]))
    assert torch.equal(xd.cpu(), g["ddpm_x_prev"])

    # Alignment contract of the stand-alone update kernels (include/edtts.h): float4 accesses only when n_per_batch % 4 == 0 AND
    # every tensor is 16-byte aligned; anything else takes the scalar path with the same results.  Contiguous tensors that start
    # 4 bytes into an allocation, and a row length that is not a multiple of 4 (odd rows then start misaligned).
    def off1(t):
        buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=DEV)
        v = buf[1:].view(t.shape)
        v.copy_(t)
        assert v.is_contiguous() and v.data_ptr() % 16 == 4
        return v

    xp, x0 = sch.get_ddim_step(off1(g["x"]), cu(g["t"]), cu(g["t_prev"]), off1(g["eps"]), eta=0.0)
    assert torch.equal(xp.cpu(), g["ddim_x_prev"]) and torch.equal(x0.cpu(), g["ddim_x0"])
    xd = sch.ddpm_step(off1(g["x"]), cu(g["t"]), cu(g["eps"]), noise=off1(g["ddpm_noise"]))
    assert torch.equal(xd.cpu(), g["ddpm_x_prev"])
    B = g["x"].shape[0]
    cut = lambda t: t.reshape(B, -1)[:, :397].contiguous()  # noqa: E731  (397 % 4 == 1)
    xp, x0 = sch.get_ddim_step(cu(cut(g["x"])), cu(g["t"]), cu(g["t_prev"]), cu(cut(g["eps"])), eta=0.0)
    assert torch.equal(xp.cpu(), cut(g["ddim_x_prev"])) and torch.equal(x0.cpu(), cut(g["ddim_x0"]))
    xd = sch.ddpm_step(cu(cut(g["x"])), cu(g["t"]), cu(cut(g["eps"])), noise=cu(cut(g["ddpm_noise"])))
    assert torch.equal(xd.cpu(), cut(g["ddpm_x_prev"]))


def test_alpha_bar_table_matches_golden(golden):
    """The one table the hot path reads, built by this host's CPU, equals the reference's (bit-exact)."""
    assert torch.equal(DiffusionSchedule(1000).alpha_bar, golden("schedule_tables")["alpha_bar"])


def _ddim_exact_fp32(alpha_bar, x, t, t_prev, eps):
    """schedule.py:179-200 with EVERY fp32 operation correctly rounded, independent of the host's libm / vector units: each
    operation is evaluated in fp64 on fp32 operands and rounded to fp32 once (for +, -, *, /, sqrt of fp32 values the fp64
    result rounded to fp32 IS the IEEE fp32 result: 53 >= 2*24 + 2).  torch.sqrt on some host CPUs is not correctly rounded
    for every input, which is why the plain fp32 oracle is only compared to 1 ulp below."""
    f32 = lambda v: v.to(torch.float32).to(torch.float64)  # noqa: E731  (round to fp32, keep carrying fp64)
    ab = alpha_bar[t].double()[:, None, None]
    abp = torch.where((t_prev >= 0), alpha_bar[t_prev.clamp(min=0)].double(), torch.ones_like(t_prev, dtype=torch.float64))[:, None, None]
    xd, ed = x.double(), eps.double()
    s1m = f32(torch.sqrt(f32(1.0 - ab)))
    sab = f32(torch.sqrt(ab))
    x0 = f32(f32(xd - f32(s1m * ed)) / sab).clamp(-3.0, 3.0)
    cdir = f32(torch.sqrt(f32(f32(1.0 - abp) - 0.0)))          # eta = 0: sigma^2 = 0
    xp = f32(f32(f32(torch.sqrt(abp)) * x0) + f32(cdir * ed))
    return xp.float(), x0.float()


def test_ddim_large_vs_oracle():
    """Full-size DDIM update (schedule.py:157-202): BIT-EQUAL to the exactly rounded fp32 evaluation of the reference's operation
    sequence, and within 1 ulp of the plain fp32 oracle run on this box's host CPU (whose torch.sqrt need not be correctly
    rounded; the committed golden vectors, made in the build container, are matched bit-exactly in test_ddim_ddpm_bit_exact)."""
    gen = torch.Generator().manual_seed(1)
    B, T, M = 16, 512, 80
    x, eps = torch.randn(B, T, M, generator=gen) * 2, torch.randn(B, T, M, generator=gen)
    t = torch.randint(0, 1000, (B,), generator=gen)
    tp = (t - 250).clamp(min=-1)
    sch = DiffusionSchedule(1000).to(DEV)
    xp, x0 = sch.get_ddim_step(cu(x), cu(t), cu(tp), cu(eps))
    ab = O.schedule_tables(1000)["alpha_bar"]
    exp_xp, exp_x0 = _ddim_exact_fp32(ab, x, t, tp, eps)
    assert torch.equal(x0.cpu(), exp_x0) and torch.equal(xp.cpu(), exp_xp)
    rxp, rx0 = O.ddim_step(ab, x, t, tp, eps)
    torch.testing.assert_close(xp.cpu(), rxp, rtol=1e-6, atol=1e-6)   # SURVEY.md section 8d (ii); values are O(1)
    torch.testing.assert_close(x0.cpu(), rx0, rtol=1e-6, atol=1e-6)
    n_diff = int((x0.cpu() != rx0).sum() + (xp.cpu() != rxp).sum())
    print(f"ddim full size: kernel == exactly-rounded fp32 (bitwise); host fp32 oracle differs from it in {n_diff} of {2 * x.numel()} values")
    assert float(x0.abs().max()) <= 3.0


def test_tiny_forward(golden):
    g = golden("tiny_ops")
    cfg = CFG(hidden=32, heads=2, layers=1, attn_window_size=4, device=DEV)
    dec = make_decoder(cfg, 3)
    e = dec(cu(g["x_t"]), cu(g["t"]), cu(g["sem_idx"]), cu(g["step_idx"])).cpu()
    assert max_abs(e, g["forward"]) < FWD_TOL
    e = dec(cu(g["x_t"]), cu(g["t"]), cu(g["sem_idx"]), None).cpu()
    assert max_abs(e, g["forward_nostep"]) < FWD_TOL
    e = dec(cu(g["x_t"]), cu(g["t"]), None, cu(g["step_idx"]), cu(g["sem_features"])).cpu()
    assert max_abs(e, g["forward_feat"]) < FWD_TOL


def test_forward_cfg_dims(golden):
    g = golden("forward_cfg")
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    e = dec(cu(g["x_t"]), cu(g["t"]), cu(g["sem_idx"]), cu(g["step_idx"])).cpu()
    assert max_abs(e, g["eps"]) < FWD_TOL
    assert max_abs(dec(cu(g["x_t"]), cu(g["t"]), cu(g["sem_idx"]), None).cpu(), g["eps_nostep"]) < FWD_TOL
    assert max_abs(dec(cu(g["x_t"]), cu(g["t"]), None, cu(g["step_idx"]), cu(g["sem_features"])).cpu(), g["eps_feat"]) < FWD_TOL
    # ragged lengths: T = 75 (not a multiple of the 32-frame tile), S = 37
    e = dec(cu(g["x_t_ragged"]), torch.tensor([321], device=DEV), cu(g["sem_idx_ragged"]), torch.tensor([2], device=DEV)).cpu()
    assert max_abs(e, g["eps_ragged"]) < FWD_TOL


def test_forward_matches_fp64_arbiter(golden):
    g = golden("forward_cfg")
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    sd = synth_state_dict(cfg, 0)
    e64 = O.decoder_forward(O.cast_sd(sd, torch.float64), g["x_t"].double(), g["t"], g["sem_idx"], g["step_idx"])
    e = dec(cu(g["x_t"]), cu(g["t"]), cu(g["sem_idx"]), cu(g["step_idx"])).cpu()
    ours, theirs = max_abs(e, e64), max_abs(g["eps"], e64)
    assert ours < 2e-5, (ours, theirs)  # same order as the reference's own fp32-vs-fp64 distance


def test_forward_cfg3_shape(golden):
    g = golden("forward_cfg3")
    cfg = CFG(hidden=256, layers=8, heads=8, device=DEV)
    dec = make_decoder(cfg, 1, max_len=1024)
    e = dec(cu(g["x_t"]), cu(g["t"]), cu(g["sem_idx"]), cu(g["step_idx"])).cpu()
    assert max_abs(e, g["eps"]) < FWD_TOL


def test_missing_context_raises():
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    with pytest.raises(ValueError):
        dec(torch.zeros(1, 32, 80, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV))


def test_length_limits_raise():
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    with pytest.raises(RuntimeError):  # T > 1000 rows of pos_emb
        dec(torch.zeros(1, 1001, 80, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV), torch.zeros(1, 8, dtype=torch.long, device=DEV))
    infer = EdgeInference(cfg, DiffusionSchedule(1000).to(DEV), None, dec)
    with pytest.raises((IndexError, RuntimeError)):  # num_steps > 16 rows of step_emb
        infer.generate_mel(torch.zeros(1, 8, dtype=torch.long, device=DEV), num_steps=17)


def test_cpu_tensors_fail_loudly():
    from edge_diffusion_tts_amd.native import EdttsError
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    with pytest.raises(EdttsError):
        dec(torch.zeros(1, 32, 80), torch.zeros(1, dtype=torch.long), torch.zeros(1, 16, dtype=torch.long))
    with pytest.raises(EdttsError):
        DiffusionSchedule(1000).get_ddim_step(torch.zeros(1, 4, 80), torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.long), torch.zeros(1, 4, 80))


def amplification_band(x_T, eps0, k=4.0):
    """the elements whose FIRST-step x0 = (x_T - sqrt(1-ab) eps0) / sqrt(ab) is not (safely) clamped,
    |x_T - sqrt(1-ab) eps0| < 4.7e-5 * k with k = 4, i.e. |x0| < 3k before the clamp: there an eps rounding difference is
    amplified 1/sqrt(alpha_bar[999]) = 64171x."""
    ab = O.schedule_tables(1000)["alpha_bar"][999].double()
    return (x_T.double() - torch.sqrt(1 - ab) * eps0.double()).abs() < 3.0 * torch.sqrt(ab) * k


def e2e_check(ours, ref, band, what, ref32=None, fp64=None, tol=E2E_TOL):
    """End-to-end criterion exactly as SURVEY.md section 8d (iii) states it: max-abs <= 1e-3 over the elements OUTSIDE the
    t=999 unclamped band; the in-band count and maximum are reported next to the reference's own fp32-vs-fp64 error on the
    same elements (ref32 / fp64 given)."""
    err = (ours.double() - ref.double()).abs()
    out_band, in_band = err[~band], err[band]
    msg = (f"{what}: outside the band max {float(out_band.max()):.2e} (n>1e-3: {int((out_band > E2E_TOL).sum())} of {out_band.numel()}), "
           f"median {float(err.median()):.2e}; in band: {in_band.numel()} elements, max {float(in_band.max()) if in_band.numel() else 0.0:.2e}")
    if ref32 is not None and fp64 is not None:
        r = (ref32.double() - fp64.double()).abs()
        msg += (f" | reference fp32 vs fp64 on the same split: outside max {float(r[~band].max()):.2e}, "
                f"in band max {float(r[band].max()) if in_band.numel() else 0.0:.2e}")
    print(msg)
    assert float(out_band.max()) <= tol, msg


def test_generate_cfg1(golden):
    """BASELINE config 1: CFG() defaults, B=1, T=256, 4-step DDIM, against the reference's CPU run."""
    g = golden("generate_cfg1")
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    sch = DiffusionSchedule(cfg.diff_steps).to(DEV)
    infer = EdgeInference(cfg, sch, torch.nn.Identity(), dec)
    out = infer.generate_mel(cu(g["sem_idx"]), 4, x_T=cu(g["x_T"])).cpu()
    assert out.shape == g["out"].shape and float(out.abs().max()) <= 3.0
    band = amplification_band(g["x_T"], g["eps0"])
    sd64 = O.cast_sd(synth_state_dict(cfg, 0), torch.float64)
    out64 = O.generate_mel(sd64, O.schedule_tables(1000, torch.float64)["alpha_bar"], g["sem_idx"], g["x_T"].double(), 4)
    e2e_check(out, g["out"], band, "generate_mel cfg1 vs reference fp32", ref32=g["out"], fp64=out64)
    # Context, not the contract: against the fp64 arbiter BOTH fp32 implementations carry their own in-band errors (reference:
    # 6.9e-3), which leak to a few neighbours through the later steps' attention; ours must stay within twice the reference's own
    # out-of-band distance from fp64.
    ref_out_band = float((g["out"].double() - out64).abs()[~band].max())
    e2e_check(out, out64, band, "generate_mel cfg1 vs fp64 arbiter", ref32=g["out"], fp64=out64, tol=max(E2E_TOL, 2 * ref_out_band))
    # teacher-forced: every step fed the reference's own x_t matches to the single-forward tolerance, and the DDIM
    # update applied to the reference's eps is bit-exact
    for i, t in enumerate([999, 749, 499, 249]):
        xin = g["x_T"] if i == 0 else g[f"x_prev{i - 1}"]
        tt = torch.full((1,), t, device=DEV)
        eps = dec(cu(xin), tt, cu(g["sem_idx"]), torch.full((1,), i, device=DEV)).cpu()
        assert max_abs(eps, g[f"eps{i}"]) < FWD_TOL, i
        xp, x0 = sch.get_ddim_step(cu(xin), tt, torch.full((1,), max(t - 250, 0), device=DEV), cu(g[f"eps{i}"]))
        assert torch.equal(xp.cpu(), g[f"x_prev{i}"]) and torch.equal(x0.cpu(), g[f"x0_{i}"]), i
    # other step counts
    sd = synth_state_dict(cfg, 0)
    for n in (1, 2, 16):
        o = infer.generate_mel(cu(g["small_sem_idx"]), n, x_T=cu(g["small_x_T"])).cpu()
        tr = []
        O.generate_mel(sd, O.schedule_tables(1000)["alpha_bar"], g["small_sem_idx"], g["small_x_T"], n, trace=tr)
        e2e_check(o, g[f"small_out_n{n}"], amplification_band(g["small_x_T"], tr[0]["eps"]), f"generate_mel {n}-step vs reference fp32")


def test_generate_default_init_exact(golden):
    """Reference default init: out_proj is zero so eps == 0 and the result is pure DDIM arithmetic -> exact."""
    g = golden("generate_cfg1")
    cfg = CFG(device=DEV)
    sd = synth_state_dict(cfg, 0)
    sd["out_proj.weight"].zero_()
    sd["out_proj.bias"].zero_()
    dec = EdgeDiffusionDecoder(cfg)
    dec.load_state_dict(sd)
    dec = dec.to(DEV).eval()
    infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), torch.nn.Identity(), dec)
    out = infer.generate_mel(cu(g["sem_idx"]), 4, x_T=cu(g["x_T"])).cpu()
    assert torch.equal(out, g["out_default_init"])


def test_generate_equals_stepwise_api():
    """The fused loop (edtts_generate) == decoder.forward + schedule.get_ddim_step called step by step, bitwise."""
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    sch = DiffusionSchedule(cfg.diff_steps).to(DEV)
    infer = EdgeInference(cfg, sch, torch.nn.Identity(), dec)
    gen = torch.Generator().manual_seed(3)
    B, S = 3, 40
    sem = torch.randint(0, 512, (B, S), generator=gen).to(DEV)
    x = torch.randn(B, 2 * S, 80, generator=gen).to(DEV)
    fused = infer.generate_mel(sem, 4, x_T=x)
    stride = 250
    for i, t in enumerate([999, 749, 499, 249]):
        tt = torch.full((B,), t, device=DEV)
        eps = dec(x, tt, sem, torch.full((B,), i, device=DEV))
        x, x0 = sch.get_ddim_step(x, tt, torch.full((B,), max(t - stride, 0), device=DEV), eps)
    assert torch.equal(fused, x0)


def test_deterministic_and_batch_invariant():
    """Run twice -> bitwise equal (no races); an utterance alone == the same utterance inside a batch (no cross-batch op)."""
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), torch.nn.Identity(), dec)
    gen = torch.Generator().manual_seed(9)
    B, S = 8, 64
    sem = torch.randint(0, 512, (B, S), generator=gen).to(DEV)
    x = torch.randn(B, 2 * S, 80, generator=gen).to(DEV)
    a = infer.generate_mel(sem, 4, x_T=x)
    b = infer.generate_mel(sem, 4, x_T=x)
    assert torch.equal(a, b)
    solo = infer.generate_mel(sem[5:6].contiguous(), 4, x_T=x[5:6].contiguous())
    assert torch.equal(solo[0], a[5])


def test_full_size_config2_properties():
    """BASELINE config 2 (B=256, T=512): too big for the oracle in seconds -> size-independent properties:
    slices of the big batch equal the same utterances run in a small batch (bitwise), outputs clamped and finite."""
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), torch.nn.Identity(), dec)
    gen = torch.Generator().manual_seed(2)
    B, S = 256, 256
    sem = torch.randint(0, 512, (B, S), generator=gen).to(DEV)
    x = torch.randn(B, 2 * S, 80, generator=gen).to(DEV)
    big = infer.generate_mel(sem, 4, x_T=x)
    assert big.shape == (B, 512, 80) and bool(torch.isfinite(big).all()) and float(big.abs().max()) <= 3.0
    idx = [0, 131, 255]
    small = infer.generate_mel(sem[idx].contiguous(), 4, x_T=x[idx].contiguous())
    assert torch.equal(small, big[idx])
    # and one of them against the CPU oracle (outside the amplification band)
    sd = synth_state_dict(cfg, 0)
    tr = []
    ref = O.generate_mel(sd, O.schedule_tables(1000)["alpha_bar"], sem[131:132].cpu(), x[131:132].cpu(), 4, trace=tr)
    e2e_check(big[131:132].cpu(), ref, amplification_band(x[131:132].cpu(), tr[0]["eps"]), "config-2 utterance 131 vs oracle")


def test_dsconv(golden):
    g = golden("dsconv")
    for tag in ("a", "b"):
        ci, co, ks = g[f"{tag}_x"].shape[1], g[f"{tag}_pw"].shape[0], g[f"{tag}_dw"].shape[-1]
        m = DepthwiseSeparableConv(ci, co, kernel_size=ks)
        m.load_state_dict({"depthwise.weight": g[f"{tag}_dw"], "pointwise.weight": g[f"{tag}_pw"], "pointwise.bias": g[f"{tag}_pb"],
                           "norm.weight": g[f"{tag}_gw"], "norm.bias": g[f"{tag}_gb"]})
        m = m.to(DEV)
        assert m.groups == int(g[f"{tag}_groups"])
        y = m(cu(g[f"{tag}_x"])).cpu()
        assert max_abs(y, g[f"{tag}_y"]) < 1e-5


def test_dsconv_stride_and_both_paths(golden, monkeypatch):
    """stride 2 / 3 (layers/conv.py:33-41: T_out = (T + 2*(k//2) - k)//stride + 1) against reference goldens; the reference's shape
    class runs the fused one-kernel path (intermediate in registers), "wide" / "long" exceed it and take the three-kernel path;
    EDTTS_DSCONV_UNFUSED forces that path for the small shapes too -- both must meet the same tolerance."""
    g, g1 = golden("dsconv_stride"), golden("dsconv")

    def run(gg, tag, stride):
        ci, co, ks = gg[f"{tag}_x"].shape[1], gg[f"{tag}_pw"].shape[0], gg[f"{tag}_dw"].shape[-1]
        m = DepthwiseSeparableConv(ci, co, kernel_size=ks, stride=stride)
        m.load_state_dict({"depthwise.weight": gg[f"{tag}_dw"], "pointwise.weight": gg[f"{tag}_pw"], "pointwise.bias": gg[f"{tag}_pb"],
                           "norm.weight": gg[f"{tag}_gw"], "norm.bias": gg[f"{tag}_gb"]})
        y = m.to(DEV)(cu(gg[f"{tag}_x"])).cpu()
        assert y.shape == gg[f"{tag}_y"].shape, tag
        return max_abs(y, gg[f"{tag}_y"])

    for tag in ("s2", "s3", "wide", "long"):
        assert run(g, tag, int(g[f"{tag}_stride"])) < 1e-5, tag
    # the same small shapes through the three-kernel path (a separate process: the switch is read once per process)
    import subprocess, sys, os
    code = ("import sys; sys.path[:0] = %r; import torch, numpy as np\n"
            "from edge_diffusion_tts_amd import DepthwiseSeparableConv\n"
            "z = np.load(%r)\n"
            "for tag in ('s2', 's3'):\n"
            "    g = {k: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + '_')}\n"
            "    m = DepthwiseSeparableConv(g[tag + '_x'].shape[1], g[tag + '_pw'].shape[0], kernel_size=g[tag + '_dw'].shape[-1], stride=int(g[tag + '_stride']))\n"
            "    m.load_state_dict({'depthwise.weight': g[tag + '_dw'], 'pointwise.weight': g[tag + '_pw'], 'pointwise.bias': g[tag + '_pb'], 'norm.weight': g[tag + '_gw'], 'norm.bias': g[tag + '_gb']})\n"
            "    y = m.cuda()(g[tag + '_x'].cuda()).cpu()\n"
            "    assert float((y - g[tag + '_y']).abs().max()) < 1e-5, tag\n"
            "print('unfused ok')\n") % (sys.path[:3], os.path.join(os.path.dirname(__file__), "golden", "dsconv_stride.npz"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, EDTTS_DSCONV_UNFUSED="1"), capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0 and "unfused ok" in r.stdout, r.stderr[-1500:]
    # stride 1 goldens of round 1 through the fused path (test_dsconv) and the benchmark shape: deterministic, finite
    x = torch.randn(8, 80, 512, generator=torch.Generator().manual_seed(3)).to(DEV)
    m = DepthwiseSeparableConv(80, 160).to(DEV)
    a, b = m(x), m(x)
    assert a.shape == (8, 160, 512) and torch.equal(a, b) and bool(torch.isfinite(a).all())
    ref = O.dsconv_forward(x.cpu(), m.depthwise.weight.cpu(), m.pointwise.weight.cpu(), m.pointwise.bias.cpu(), m.norm.weight.cpu(), m.norm.bias.cpu(), m.groups)
    assert max_abs(a.cpu(), ref) < 2e-5


def test_dsconv_group_pipelined_form_vs_oracle():
    """The group-pipelined one-kernel form (C_out = 160, GroupNorm(8), stride 1 -- the layer conv.py:47-48 builds; k_dsconv_grouped)
    against the CPU oracle on shapes its frame masks and pass structure must get right: T not a multiple of 4 (scalar stores, frames
    past T_out inside a wave's rows), T below one pass, T = 1, fewer input channels than the padded 80, kernel 5, an input with a
    large offset (the per-wave centred statistics must not cancel), and non-trivial GroupNorm weights."""
    gen = torch.Generator().manual_seed(11)
    for (B, ci, T, ks, shift) in [(3, 80, 301, 3, 0.0), (2, 56, 509, 5, 0.0), (2, 80, 7, 3, 0.0), (1, 80, 1, 3, 0.0), (2, 80, 256, 5, 0.0),
                                  (2, 80, 260, 3, 50.0), (2, 17, 512, 1, 0.0)]:
        m = DepthwiseSeparableConv(ci, 160, kernel_size=ks)
        with torch.no_grad():
            m.norm.weight.copy_(torch.randn(160, generator=gen))
            m.norm.bias.copy_(torch.randn(160, generator=gen))
            m.pointwise.bias.copy_(torch.randn(160, generator=gen))
        x = torch.randn(B, ci, T, generator=gen) + shift
        ref = O.dsconv_forward(x.double(), m.depthwise.weight.double(), m.pointwise.weight.double(), m.pointwise.bias.double(),
                               m.norm.weight.double(), m.norm.bias.double(), m.groups).float()
        ref32 = O.dsconv_forward(x, m.depthwise.weight, m.pointwise.weight, m.pointwise.bias, m.norm.weight, m.norm.bias, m.groups)
        y = m.to(DEV)(x.to(DEV)).cpu()
        assert y.shape == ref.shape
        # (offset 50: z sits ~50 sigma away from zero and every fp32 evaluation loses digits there -- the bar is the fp32 oracle's own error)
        tol = max(2e-5, 3.0 * max_abs(ref32, ref))
        assert max_abs(y, ref) < tol, f"B={B} C_in={ci} T={T} k={ks} shift={shift}: {max_abs(y, ref):.3e} vs fp64, the fp32 oracle {max_abs(ref32, ref):.3e}"


def test_forward_large_score_range():
    """Forces the deferred-max rescale branch of the attention kernels (taken when a later key chunk exceeds the softmax
    reference point by > 2^32): q/k projections scaled so that score ranges span hundreds of octaves, plus a spiked context
    token.  Compared with the fp64 arbiter (saturated softmax is ill-conditioned in fp32 for BOTH implementations)."""
    cfg = CFG(device=DEV)
    sd = synth_state_dict(cfg, 0)
    for l in range(cfg.layers):
        sd[f"layers.{l}.attn.qkv.weight"][: 2 * cfg.hidden] *= 9.0      # q and k rows
        sd[f"layers.{l}.cross_attn.q_proj.weight"] *= 9.0
        sd[f"layers.{l}.cross_attn.kv_up_proj.weight"][: cfg.hidden] *= 9.0
    dec = EdgeDiffusionDecoder(cfg)
    dec.load_state_dict(sd)
    dec = dec.to(DEV).eval()
    gen = torch.Generator().manual_seed(4)
    B, T, S = 2, 160, 80
    x = torch.randn(B, T, 80, generator=gen) * 1.5
    sem = torch.randint(0, 512, (B, S), generator=gen)
    t, si = torch.tensor([700, 30]), torch.tensor([1, 2])
    e = dec(cu(x), cu(t), cu(sem), cu(si)).cpu()
    ref64 = O.decoder_forward(O.cast_sd(sd, torch.float64), x.double(), t, sem, si)
    ref32 = O.decoder_forward(sd, x, t, sem, si)
    ours, theirs = max_abs(e, ref64), max_abs(ref32, ref64)
    print(f"large-score-range forward: ours vs fp64 {ours:.2e}, oracle fp32 vs fp64 {theirs:.2e}")
    assert bool(torch.isfinite(e).all())
    assert ours < max(5e-4, 20 * theirs)


@pytest.mark.parametrize("window", [4, 9, 37])
def test_small_window_all_scores_far_below_zero(window):
    """Rows whose FIRST key chunk is fully masked (the band does not start on a 32-key chunk boundary) must take their softmax
    reference from the first chunk that has a visible key -- also when ALL their scores are hundreds of octaves below zero,
    where a reference left at 0 would underflow every exp2 and normalise the row to 0.  k = -alpha * q with a nearly constant
    input makes every score q_i . k_j = -alpha |q|^2 hugely negative."""
    cfg = CFG(device=DEV, attn_window_size=window)
    sd = synth_state_dict(cfg, 3)
    H = cfg.hidden
    for l in range(cfg.layers):
        w = sd[f"layers.{l}.attn.qkv.weight"]
        w[H:2 * H] = -6.0 * w[:H]  # k rows = -6 * q rows
        w[:H] *= 6.0
    dec = EdgeDiffusionDecoder(cfg)
    dec.load_state_dict(sd)
    dec = dec.to(DEV).eval()
    gen = torch.Generator().manual_seed(6)
    B, T, S = 2, 160, 80
    x = torch.randn(B, 1, 80, generator=gen).expand(B, T, 80) * 1.5 + 0.01 * torch.randn(B, T, 80, generator=gen)
    x = x.contiguous()
    sem = torch.randint(0, 512, (B, S), generator=gen)
    t, si = torch.tensor([650, 20]), torch.tensor([0, 3])
    e = dec(cu(x), cu(t), cu(sem), cu(si)).cpu()
    ref64 = O.decoder_forward(O.cast_sd(sd, torch.float64), x.double(), t, sem, si, window=window)
    ref32 = O.decoder_forward(sd, x, t, sem, si, window=window)
    ours, theirs = max_abs(e, ref64), max_abs(ref32, ref64)
    print(f"window {window}, all scores << 0: ours vs fp64 {ours:.2e}, oracle fp32 vs fp64 {theirs:.2e}")
    assert bool(torch.isfinite(e).all())
    assert ours < max(5e-4, 20 * theirs)


def test_forward_random_geometries():
    """Seeded sweep over (B, T, S, window): ragged lengths, windows on and off the 16-key tile grid, windows larger than the
    utterance, full attention -- the mask / chunk-order / tile-clamping logic against the oracle."""
    rng = np.random.default_rng(20260101)
    worst = 0.0
    for case in range(14):
        B = int(rng.integers(1, 4))
        S = int(rng.integers(1, 110))
        T = 2 * S
        window = [None, int(rng.integers(1, 100)), int(rng.integers(1, 20)), 64][case % 4]
        cfg = CFG(device=DEV, attn_window_size=window)
        sd = synth_state_dict(cfg, 5 + case % 3)
        dec = EdgeDiffusionDecoder(cfg)
        dec.load_state_dict(sd)
        dec = dec.to(DEV).eval()
        gen = torch.Generator().manual_seed(100 + case)
        x = torch.randn(B, T, 80, generator=gen)
        sem = torch.randint(0, 512, (B, S), generator=gen)
        t = torch.randint(0, 1000, (B,), generator=gen)
        si = torch.randint(0, 16, (B,), generator=gen)
        e = dec(cu(x), cu(t), cu(sem), cu(si)).cpu()
        err = max_abs(e, O.decoder_forward(sd, x, t, sem, si, window=window))
        worst = max(worst, err)
        assert err < FWD_TOL, (case, B, T, S, window, err)
    print(f"random geometries: worst max-abs {worst:.2e}")


@pytest.mark.parametrize("B,S", [(1, 1), (3, 7), (5, 16), (2, 100)])
def test_forward_small_and_odd_shapes(B, S):
    """Shortest possible utterance (S=1 -> T=2), lengths that are not multiples of any tile, odd batches."""
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    sd = synth_state_dict(cfg, 0)
    gen = torch.Generator().manual_seed(100 + S)
    T = 2 * S
    x = torch.randn(B, T, 80, generator=gen)
    sem = torch.randint(0, 512, (B, S), generator=gen)
    t = torch.randint(0, 1000, (B,), generator=gen)
    si = torch.randint(0, 16, (B,), generator=gen)
    e = dec(cu(x), cu(t), cu(sem), cu(si)).cpu()
    assert max_abs(e, O.decoder_forward(sd, x, t, sem, si)) < FWD_TOL


def test_forward_maximum_lengths():
    """T = 1000 and S = 500 (the reference's positional tables hold 1000 / 512 rows), max token id, max step index."""
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    sd = synth_state_dict(cfg, 0)
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(1, 1000, 80, generator=gen)
    sem = torch.randint(0, 512, (1, 500), generator=gen)
    sem[0, :3] = 511
    t, si = torch.tensor([999]), torch.tensor([15])
    e = dec(cu(x), cu(t), cu(sem), cu(si)).cpu()
    assert max_abs(e, O.decoder_forward(sd, x, t, sem, si)) < FWD_TOL
    # S = 512 is the context table's limit; T stays below 1000 by passing a shorter x than 2*S would give
    sem2 = torch.randint(0, 512, (1, 512), generator=gen)
    x2 = torch.randn(1, 96, 80, generator=gen)
    e2 = dec(cu(x2), cu(t), cu(sem2), cu(si)).cpu()
    assert max_abs(e2, O.decoder_forward(sd, x2, t, sem2, si)) < FWD_TOL


def test_forward_more_than_32_key_chunks():
    """The per-wave interior-chunk bit mask covers the first 32 chunks (1 024 keys: beyond the reference's own length limits); chunks
    past it take the always-correct masked path.  Full attention over T = 1 100 frames (35 chunks) and a window of 600."""
    for window in (None, 600):
        cfg = CFG(hidden=32, heads=2, layers=1, attn_window_size=window, device=DEV)
        sd = synth_state_dict(cfg, 3, max_pos=1200)
        dec = EdgeDiffusionDecoder(cfg, max_len=1200)
        dec.load_state_dict(sd)
        dec = dec.to(DEV).eval()
        gen = torch.Generator().manual_seed(44)
        x = torch.randn(1, 1100, 80, generator=gen)
        sem = torch.randint(0, 512, (1, 64), generator=gen)
        t, si = torch.tensor([300]), torch.tensor([2])
        e = dec(cu(x), cu(t), cu(sem), cu(si)).cpu()
        ref = O.decoder_forward(sd, x, t, sem, si, heads=2, window=window)
        assert max_abs(e, ref) < FWD_TOL, (window, max_abs(e, ref))


def test_forward_full_attention_and_other_windows():
    """attn_window_size=None (full self-attention, layers/attention.py:94) and windows that are not multiples of the key tile."""
    for window in (None, 5, 37, 200):
        cfg = CFG(device=DEV, attn_window_size=window)
        dec = make_decoder(cfg, 2)
        sd = synth_state_dict(cfg, 2)
        gen = torch.Generator().manual_seed(5)
        B, S = 2, 60
        x = torch.randn(B, 2 * S, 80, generator=gen)
        sem = torch.randint(0, 512, (B, S), generator=gen)
        t, si = torch.tensor([400, 3]), torch.tensor([2, 9])
        e = dec(cu(x), cu(t), cu(sem), cu(si)).cpu()
        assert max_abs(e, O.decoder_forward(sd, x, t, sem, si, window=window)) < FWD_TOL, window


def test_weights_repacked_after_update():
    """Parameter changes (load_state_dict / in-place edits) must reach the packed blob."""
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(1, 64, 80, generator=gen)
    sem = torch.randint(0, 512, (1, 32), generator=gen)
    t, si = torch.tensor([500]), torch.tensor([1])
    e0 = dec(cu(x), cu(t), cu(sem), cu(si)).cpu()
    sd1 = synth_state_dict(cfg, 1)
    dec.load_state_dict(sd1)
    e1 = dec(cu(x), cu(t), cu(sem), cu(si)).cpu()
    assert max_abs(e1, O.decoder_forward(sd1, x, t, sem, si)) < FWD_TOL and max_abs(e0, e1) > 1e-2
    with torch.no_grad():
        dec.out_proj.bias.add_(1.0)
    e2 = dec(cu(x), cu(t), cu(sem), cu(si)).cpu()
    assert max_abs(e2, e1 + 1.0) < 1e-5


def _ddpm_setup(diff_steps, B, S, seed):
    cfg = CFG(device=DEV, diff_steps=diff_steps)
    dec = make_decoder(cfg, 0)
    sch = DiffusionSchedule(cfg.diff_steps).to(DEV)
    infer = EdgeInference(cfg, sch, torch.nn.Identity(), dec)
    gen = torch.Generator().manual_seed(seed)
    sem = torch.randint(0, 512, (B, S), generator=gen)
    x_T = torch.randn(B, 2 * S, 80, generator=gen)
    return cfg, infer, sem, x_T, gen


def test_sample_ddpm_injected_noise_vs_oracle():
    """BASELINE config 5's loop (decoder with step_idx=None + ddpm_step), short schedule, per-step noise injected so the
    CPU oracle sees the same draws.  Early steps multiply by 1/sqrt(alpha) ~ 100, so the
    comparison is relative to the state's magnitude."""
    cfg, infer, sem, x_T, gen = _ddpm_setup(50, 2, 24, 3)
    n = 6
    noise = torch.randn(n, 2, 48, 80, generator=gen)
    out = infer.sample_ddpm(cu(sem), n, x_T=cu(x_T), noise=cu(noise)).cpu()
    ref = O.sample_ddpm(synth_state_dict(cfg, 0), O.schedule_tables(50), sem, x_T, noise, n)
    scale = float(ref.abs().max())
    assert bool(torch.isfinite(out).all()) and max_abs(out, ref) < 2e-4 * max(scale, 1.0), (max_abs(out, ref), scale)
    # one step == public API pieces: decoder.forward + DiffusionSchedule.ddpm_step with the same noise (bitwise)
    one = infer.sample_ddpm(cu(sem), 1, x_T=cu(x_T), noise=cu(noise[:1]))
    tt = torch.full((2,), 49, device=DEV)
    eps = infer.decoder(cu(x_T), tt, cu(sem), None)
    step = infer.schedule.ddpm_step(cu(x_T), tt, eps, noise=cu(noise[0]))
    assert torch.equal(one, step)


def test_sample_ddpm_philox_noise():
    """In-kernel Philox noise: deterministic per seed, different across seeds / steps, standard-normal moments, and the last
    step (t = 0) adds no noise (schedule.py:236)."""
    cfg, infer, sem, x_T, _ = _ddpm_setup(20, 4, 64, 5)
    a = infer.sample_ddpm(cu(sem), 3, x_T=cu(x_T), seed=7)
    b = infer.sample_ddpm(cu(sem), 3, x_T=cu(x_T), seed=7)
    c = infer.sample_ddpm(cu(sem), 3, x_T=cu(x_T), seed=8)
    assert torch.equal(a, b) and not torch.equal(a, c) and bool(torch.isfinite(a).all())
    # recover the noise of one step: x_1step(seed) - x_1step(zero noise) = sd * n
    z = infer.sample_ddpm(cu(sem), 1, x_T=cu(x_T), noise=torch.zeros(1, 4, 128, 80, device=DEV))
    p = infer.sample_ddpm(cu(sem), 1, x_T=cu(x_T), seed=11)
    sd_ = infer.schedule.ddpm_coefficients(19)[2]
    nz = ((p - z) / sd_).flatten().double().cpu()
    assert abs(float(nz.mean())) < 0.02 and abs(float(nz.std()) - 1.0) < 0.02
    assert abs(float((nz ** 3).mean())) < 0.05 and abs(float((nz ** 4).mean()) - 3.0) < 0.15
    # full schedule down to t = 0 runs and stays finite
    full = infer.sample_ddpm(cu(sem[:1]), None, x_T=cu(x_T[:1]), seed=1)
    assert full.shape == (1, 128, 80) and bool(torch.isfinite(full).all())


def test_samplers_are_graph_capturable():
    """The C ABI never allocates or synchronises and takes per-step scalars by value, so a whole generate_mel (23 launches)
    or DDPM loop can be captured into a hipGraph and replayed on new inputs written into the static buffers."""
    cfg, infer, sem, x_T, gen = _ddpm_setup(1000, 3, 48, 9)
    sem_s, x_s = cu(sem).clone(), cu(x_T).clone()
    eager = infer.generate_mel(sem_s, 4, x_T=x_s)          # warm-up: packs weights, sizes the workspace, sets attributes
    eager_d = infer.sample_ddpm(sem_s, 8, x_T=x_s, seed=3)
    torch.cuda.synchronize()
    g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        out_g = infer.generate_mel(sem_s, 4, x_T=x_s)
    with torch.cuda.graph(g2):
        out_d = infer.sample_ddpm(sem_s, 8, x_T=x_s, seed=3)
    g1.replay(); g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_g, eager) and torch.equal(out_d, eager_d)
    # new inputs through the same static buffers
    sem2 = torch.randint(0, 512, tuple(sem.shape), generator=gen)
    x2 = torch.randn(tuple(x_T.shape), generator=gen)
    sem_s.copy_(cu(sem2)); x_s.copy_(cu(x2))
    g1.replay(); g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_g, infer.generate_mel(cu(sem2), 4, x_T=cu(x2)))
    assert torch.equal(out_d, infer.sample_ddpm(cu(sem2), 8, x_T=cu(x2), seed=3))


def test_dpm_solver_pp(golden):
    """DPM-Solver++ sampler through edtts_sample_multistep, against the reference's CPU run:
    orders 1-3, 4 and 7 steps, final sample and every intermediate clamped x0; x0-prediction mode; token conditioning."""
    from edge_diffusion_tts_amd import DPMSolverPP
    g = golden("dpmpp")
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    sch = DiffusionSchedule(cfg.diff_steps).to(DEV)
    for order in (1, 2, 3):
        solver = DPMSolverPP(sch, order=order)
        for n in (4, 7):
            assert solver.get_time_steps(n, 950).tolist() == g[f"ts_o{order}_n{n}"].tolist()
            out, inter = solver.sample(dec, cu(g["x_T"]), cu(g["sem_features"]), num_steps=n, return_intermediates=True)
            assert max_abs(out.cpu(), g[f"out_o{order}_n{n}"]) < FWD_TOL, (order, n)
            assert max_abs(torch.stack(inter).cpu(), g[f"x0s_o{order}_n{n}"]) < FWD_TOL, (order, n)
    out = DPMSolverPP(sch, order=2, predict_x0=True).sample(dec, cu(g["x_T"]), cu(g["sem_features"]), num_steps=5)
    assert max_abs(out.cpu(), g["out_px0_o2_n5"]) < FWD_TOL
    assert DPMSolverPP(sch).get_time_steps(10).tolist() == g["ts_default_n10"].tolist()
    # fused call == the public step-by-step pieces (decoder.forward + the update methods), to rounding
    solver = DPMSolverPP(sch, order=2)
    ts = solver.get_time_steps(4, 950)
    x = cu(g["x_T"])
    B = x.shape[0]
    hist, th = [], []
    for i, t in enumerate(ts.tolist()):
        tt = torch.full((B,), t, device=DEV)
        v = dec(x, tt, None, torch.full((B,), i, device=DEV), cu(g["sem_features"]))
        x0 = solver.model_to_x0(v, x, tt).clamp(-3, 3)
        tp = torch.full((B,), ts[i + 1].item() if i < 3 else 0, device=DEV)
        x = solver.first_order_update(x, x0, tt, tp) if not hist else solver.second_order_update(x, x0, hist[-1], tt, tp, th[-1])
        hist.append(x0); th.append(tp)
    fused = solver.sample(dec, cu(g["x_T"]), cu(g["sem_features"]), num_steps=4)
    assert max_abs(fused, x) < 2e-5
    with pytest.raises((IndexError, RuntimeError)):
        solver.sample(dec, cu(g["x_T"]), cu(g["sem_features"]), num_steps=17)


def test_no_cross_block_hazard_when_utterances_straddle_block_rounds():
    """Regression test for a write-after-read hazard between blocks of ONE layer launch: the QKV tail writes the next layer's
    K / V^T while later-scheduled neighbouring blocks still need this layer's halo rows.  T = 768 gives 6 blocks per utterance
    and B = 64 gives 384 blocks; with the XCD-aware remap each XCD owns 48 consecutive tiles of which 32 are resident first, so
    the utterances covering tiles x*48+31 | x*48+32 (utterances 5, 13, 21, ...) have blocks in both batches.  q / k / v^T
    therefore ping-pong between two buffer sets.  Every utterance of the big batch must equal the same utterance run alone."""
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    gen = torch.Generator().manual_seed(21)
    B, S = 64, 384
    x = torch.randn(B, 2 * S, 80, generator=gen).to(DEV)
    sem = torch.randint(0, 512, (B, S), generator=gen).to(DEV)
    t = torch.full((B,), 600, device=DEV)
    si = torch.full((B,), 1, device=DEV)
    big = dec(x, t, sem, si)
    for u in (0, 5, 13, 21, 29, 37, 45, 53, 61, 63):
        solo = dec(x[u:u + 1].contiguous(), t[:1], sem[u:u + 1].contiguous(), si[:1])
        assert torch.equal(solo[0], big[u]), u


def test_forward_without_adaln(golden):
    """CFG(use_adaln=False): plain RMSNorm blocks (layers/transformer.py:101-104,119-122,142-157) against the reference's output."""
    g = golden("forward_noadaln")
    cfg = CFG(use_adaln=False, device=DEV)
    dec = make_decoder(cfg, 4)
    e = dec(cu(g["x_t"]), cu(g["t"]), cu(g["sem_idx"]), cu(g["step_idx"])).cpu()
    assert max_abs(e, g["eps"]) < FWD_TOL
    # and it is a different network from the AdaLN one with the same shared weights (the branch is really taken)
    assert float(g["eps"].abs().max()) > 0.1


def test_checkpoint_interop_on_device(golden, tmp_path):
    """SURVEY.md section 8f-2 on the HIP path: a reference-style checkpoint (train.py:291-297) whose decoder keys carry the
    `_orig_mod.` prefix torch.compile leaves (train.py:84-86,195), with the FSQ-sized 2304-row token embedding (train_v2.py:246)
    while the stored cfg still says codebook_size=512 -> from_checkpoint -> eps equal to the reference's output, ids up to 2303."""
    g = golden("forward_fsq")
    cfg = CFG(codebook_size=2304, device="cpu")
    sd = synth_state_dict(cfg, 6)
    stored_cfg = CFG(device="cpu").to_dict()                   # codebook_size = 512 in the stored config
    ck = {"decoder": {"_orig_mod." + k: v.clone() for k, v in sd.items()}, "cfg": stored_cfg, "encoder_proj": {}, "encoder_vq": {}}
    path = tmp_path / "edge_model_final.pt"
    torch.save(ck, path)
    dec = EdgeDiffusionDecoder.from_checkpoint(str(path), device=DEV)
    assert dec.cfg.codebook_size == 2304 and int(g["sem_idx"].max()) == 2303
    e = dec(cu(g["x_t"]), cu(g["t"]), cu(g["sem_idx"]), cu(g["step_idx"])).cpu()
    assert max_abs(e, g["eps"]) < FWD_TOL
    assert max_abs(e, O.decoder_forward(sd, g["x_t"], g["t"], g["sem_idx"], g["step_idx"])) < FWD_TOL
    # the sampler runs on the loaded decoder too
    infer = EdgeInference(dec.cfg, DiffusionSchedule(1000).to(DEV), torch.nn.Identity(), dec)
    out = infer.generate_mel(cu(g["sem_idx"]), 2, x_T=cu(g["x_t"]))
    assert out.shape == g["x_t"].shape and bool(torch.isfinite(out).all())


class _StubEncoder(torch.nn.Module):
    """Stands in for SemanticEncoder (HuBERT + quantiser, models/encoder.py -- a network fetch, out of scope): returns the
    reference's 5-tuple (z_q, indices, commit_loss, perplexity, features) with deterministic indices derived from the waveform."""

    def __init__(self, codebook_size, hop=320):
        super().__init__()
        self.codebook_size, self.hop, self.calls = codebook_size, hop, 0

    def forward(self, wav):
        self.calls += 1
        S = wav.shape[1] // self.hop
        frames = wav[:, :S * self.hop].reshape(wav.shape[0], S, self.hop)
        idx = (frames.abs().mean(-1) * 7919.0).long() % self.codebook_size
        return None, idx, torch.zeros(()), torch.zeros(()), None


def test_generate_from_audio_with_stub_encoder():
    """EdgeInference.generate_from_audio (inference.py:55-62): wav -> encoder(wav)[1] -> generate_mel; 1-D wav is unsqueezed."""
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    enc = _StubEncoder(cfg.codebook_size).to(DEV)
    infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), enc, dec)
    gen = torch.Generator().manual_seed(12)
    wav = torch.randn(2, 320 * 20, generator=gen)
    torch.manual_seed(99)
    out = infer.generate_from_audio(wav, num_steps=4)
    assert out.shape == (2, 40, cfg.n_mels) and enc.calls == 1 and float(out.abs().max()) <= 3.0
    # same thing by hand: the encoder's indices through generate_mel with the same device RNG state
    _, idx, _, _, _ = enc(wav.to(DEV))
    torch.manual_seed(99)
    assert torch.equal(out, infer.generate_mel(idx, 4))
    one = infer.generate_from_audio(wav[0], num_steps=1)     # 1-D waveform
    assert one.shape == (1, 40, cfg.n_mels)


def test_out_of_range_indices_are_flagged():
    """Where the reference raises IndexError (token id >= codebook_size, step_idx >= 16) the kernels clamp and record it; the
    debug mode EDTTS_CHECK_INDICES=1 turns the record into an IndexError (include/edtts.h: edtts_index_errors)."""
    from edge_diffusion_tts_amd import native
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    x = torch.zeros(1, 32, 80, device=DEV)
    t = torch.tensor([5], device=DEV)
    good = torch.zeros(1, 16, dtype=torch.long, device=DEV)
    dec(x, t, good, torch.tensor([3], device=DEV))
    ws = dec.workspace(1, 32, 16, 1, x.device)
    assert native.index_errors(ws) == 0
    bad = good.clone()
    bad[0, 3] = 512
    dec(x, t, bad, torch.tensor([16], device=DEV))
    assert native.index_errors(ws) == 3 and native.index_errors(ws) == 0   # both bits; reading clears
    old = native.CHECK_INDICES
    native.CHECK_INDICES = True
    try:
        with pytest.raises(IndexError, match="sem_idx"):
            dec(x, t, bad, None)
        with pytest.raises(IndexError, match="step_idx"):
            dec(x, t, good, torch.tensor([-1], device=DEV))
        dec(x, t, good, torch.tensor([15], device=DEV))
        with pytest.raises(IndexError):
            DiffusionSchedule(1000).to(DEV).get_ddim_step(x, torch.tensor([1000], device=DEV), torch.tensor([0], device=DEV), x)
    finally:
        native.CHECK_INDICES = old


def test_library_noise_is_shard_invariant():
    """edtts_randn / the in-kernel DDPM noise are keyed by the GLOBAL element index: a rank that owns rows [lo, hi) draws exactly
    the values a single GPU draws for those rows (no rank has to materialise the global noise)."""
    from edge_diffusion_tts_amd import native
    full = native.randn((6, 64, 80), DEV, seed=5)
    part = native.randn((2, 64, 80), DEV, seed=5, elem_offset=3 * 64 * 80)
    assert torch.equal(part, full[3:5])
    z = native.randn((64, 512, 80), DEV, seed=1).flatten().double()
    assert abs(float(z.mean())) < 3e-3 and abs(float(z.std()) - 1.0) < 3e-3 and abs(float((z ** 4).mean()) - 3.0) < 0.03
    assert not torch.equal(native.randn((2, 64, 80), DEV, seed=6), full[:2])
    assert torch.equal(native.randn((2, 64, 80), DEV, seed=5, scale=0.5), full[:2] * 0.5)
    cfg, infer, sem, x_T, _ = _ddpm_setup(20, 4, 32, 5)
    whole = infer.sample_ddpm(cu(sem), 3, x_T=cu(x_T), seed=7)
    shard = infer.sample_ddpm(cu(sem[2:]), 3, x_T=cu(x_T[2:]), seed=7, batch_offset=2)
    assert torch.equal(shard, whole[2:])
    # generate_mel(seed=...) draws its start noise the same way
    a = infer.generate_mel(cu(sem), 4, seed=3)
    b = infer.generate_mel(cu(sem[1:3]), 4, seed=3, batch_offset=1)
    assert torch.equal(b, a[1:3])


def _empty_shard_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    from edge_diffusion_tts_amd.parallel import ShardedEdgeInference
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.chdir("/tmp")
    dist.init_process_group("gloo", rank=rank, world_size=world)  # rehearsal backend: both ranks share this box's one GPU
    try:
        cfg = CFG(device=DEV)
        dec = make_decoder(cfg, 0)
        infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), torch.nn.Identity(), dec)
        sem = torch.randint(0, 512, (1, 32), generator=torch.Generator().manual_seed(4)).to(DEV)  # ONE utterance, two ranks
        out = ShardedEdgeInference(infer).generate_mel(sem, 4, seed=9)   # library Philox noise: the CUDA branch of the sharded path
        ref = infer.generate_mel(sem, 4, seed=9)
        sem3 = torch.randint(0, 512, (3, 32), generator=torch.Generator().manual_seed(5)).to(DEV)  # ragged: 2 + 1
        out3 = ShardedEdgeInference(infer).generate_mel(sem3, 4, seed=9)
        q.put((rank, bool(torch.equal(out, ref)) and bool(torch.equal(out3, infer.generate_mel(sem3, 4, seed=9)))))
    finally:
        dist.destroy_process_group()


def test_sharded_sampler_with_fewer_utterances_than_ranks():
    """ShardedEdgeInference on the GPU path with total < world (BASELINE config 1, B = 1, on 2 ranks): the rank with the empty shard
    draws nothing (a zero-element tensor has a NULL data pointer: edtts_randn must not be called with it), skips the sampler and
    joins the gather; every rank receives the single-GPU result.  Two processes share the one GPU of this box, gloo collective."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_empty_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    res = dict(q.get(timeout=5) for _ in procs)
    assert all(p.exitcode == 0 for p in procs) and res == {0: True, 1: True}
    # the entry point itself: n = 0 is a no-op even with a NULL pointer; reserved stream ids are refused
    from edge_diffusion_tts_amd import native
    assert native.randn((0, 64, 80), DEV, seed=1).shape == (0, 64, 80)
    assert native.lib().edtts_randn(None, 0, 1, 0, 0, 1.0, None) == 0
    with pytest.raises(native.EdttsError, match="reserved"):
        native.randn((1, 4, 80), DEV, seed=1, stream_id=0x10000)


def test_config5_full_size_graph():
    """BASELINE config 5 at full size: 1000-step DDPM sampler, B=64, T=512, captured as ONE hipGraph (5 launches per step) and
    replayed; the replay equals the eager run bitwise and stays finite."""
    cfg, infer, sem, x_T, _ = _ddpm_setup(1000, 64, 256, 17)
    sem_s, x_s = cu(sem), cu(x_T)
    eager = infer.sample_ddpm(sem_s, None, x_T=x_s, seed=2)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = infer.sample_ddpm(sem_s, None, x_T=x_s, seed=2)
    g.replay()
    torch.cuda.synchronize()
    assert out.shape == (64, 512, 80) and bool(torch.isfinite(out).all())
    assert torch.equal(out, eager)


# ---------------------------------------------------------------------------------------------------------------
# bf16 instance (BASELINE config 3): contractions on bf16 MFMA, fp32 accumulate / residual / norms / softmax.
# Tolerance: the reference's OWN bf16 path (torch.autocast("cpu", bfloat16) -- its AMP precedent, utils/speed_utils.py:70,
# train_v2.py:290) differs from its fp32 output by max 1.23e-2 / rms 2.54e-3 on the config-3 golden (tests/golden/
# make_golden_r2.py prints it; eps has rms 0.60).  The bf16 kernels must be at least as close to the reference's fp32 output
# as the reference's own bf16 run is: rms <= 2.6e-3 and max <= 2e-2 (bf16 has 8 significant bits: relative rounding 2^-9 = 2e-3).
# ---------------------------------------------------------------------------------------------------------------
BF16_RMS_TOL, BF16_MAX_TOL = 2.6e-3, 2e-2


def rms(a, b):
    return float((a.double() - b.double()).pow(2).mean().sqrt())


@pytest.mark.parametrize("B,S,window", [(2, 40, 64), (1, 77, 9), (3, 16, None), (2, 5, 3)])
def test_bf16_tiny_forward(B, S, window):
    """hidden=64, heads=2 (head_dim 32), 2 layers: ragged lengths, windows on and off the tile grid, full attention."""
    cfg = CFG(hidden=64, heads=2, layers=2, attn_window_size=window, device=DEV)
    sd = synth_state_dict(cfg, 7)
    dec = EdgeDiffusionDecoder(cfg, compute_dtype="bf16")
    dec.load_state_dict(sd)
    dec = dec.to(DEV).eval()
    gen = torch.Generator().manual_seed(10 * B + S)
    x = torch.randn(B, 2 * S, 80, generator=gen)
    sem = torch.randint(0, 512, (B, S), generator=gen)
    t = torch.randint(0, 1000, (B,), generator=gen)
    si = torch.randint(0, 16, (B,), generator=gen)
    e = dec(cu(x), cu(t), cu(sem), cu(si)).cpu()
    ref = O.decoder_forward(sd, x, t, sem, si, heads=2, window=window)
    assert bool(torch.isfinite(e).all())
    assert rms(e, ref) < BF16_RMS_TOL and max_abs(e, ref) < BF16_MAX_TOL, (rms(e, ref), max_abs(e, ref))
    # sem_features path and step_idx=None through the same kernels
    feats = torch.randn(B, S, cfg.semantic_dim, generator=gen)
    e2 = dec(cu(x), cu(t), None, None, cu(feats)).cpu()
    ref2 = O.decoder_forward(sd, x, t, None, None, feats, heads=2, window=window)
    assert rms(e2, ref2) < BF16_RMS_TOL and max_abs(e2, ref2) < BF16_MAX_TOL


def test_bf16_cfg3_shape(golden):
    """BASELINE config 3's shape (hidden=256, L=8, heads=8, T=1024, S=512) against the reference's fp32 output, next to the
    reference's own autocast(bf16) output on the same input."""
    g, ga = golden("forward_cfg3"), golden("forward_cfg3_bf16")
    cfg = CFG(hidden=256, layers=8, heads=8, device=DEV)
    dec = EdgeDiffusionDecoder(cfg, max_len=1024, compute_dtype="bf16")
    dec.load_state_dict(synth_state_dict(cfg, 1, max_pos=1024))
    dec = dec.to(DEV).eval()
    e = dec(cu(g["x_t"]), cu(g["t"]), cu(g["sem_idx"]), cu(g["step_idx"])).cpu()
    ours, theirs = (rms(e, g["eps"]), max_abs(e, g["eps"])), (rms(ga["eps_autocast"], g["eps"]), max_abs(ga["eps_autocast"], g["eps"]))
    print(f"config-3 shape, vs the reference's fp32 output: ours bf16 rms {ours[0]:.2e} max {ours[1]:.2e}; "
          f"reference autocast(bf16) rms {theirs[0]:.2e} max {theirs[1]:.2e}")
    assert ours[0] < BF16_RMS_TOL and ours[1] < BF16_MAX_TOL
    assert ours[0] <= theirs[0] * 1.05  # at least as close to fp32 as the reference's own bf16 run


def test_bf16_sampler_properties():
    """4-step DDIM through the bf16 kernels: deterministic, batch-invariant (bitwise), fused == stepwise (bitwise), clamped, and
    close to the fp32 instance's result outside the t=999 band."""
    cfg = CFG(hidden=64, heads=2, layers=2, device=DEV)
    sd = synth_state_dict(cfg, 7)
    d16 = EdgeDiffusionDecoder(cfg, compute_dtype="bf16")
    d16.load_state_dict(sd)
    d16 = d16.to(DEV).eval()
    sch = DiffusionSchedule(cfg.diff_steps).to(DEV)
    infer = EdgeInference(cfg, sch, torch.nn.Identity(), d16)
    gen = torch.Generator().manual_seed(31)
    B, S = 6, 48
    sem = torch.randint(0, 512, (B, S), generator=gen).to(DEV)
    x = torch.randn(B, 2 * S, 80, generator=gen).to(DEV)
    a = infer.generate_mel(sem, 4, x_T=x)
    assert torch.equal(a, infer.generate_mel(sem, 4, x_T=x)) and float(a.abs().max()) <= 3.0
    assert torch.equal(infer.generate_mel(sem[2:3].contiguous(), 4, x_T=x[2:3].contiguous())[0], a[2])
    xs, x0 = x, None
    for i, t in enumerate([999, 749, 499, 249]):
        tt = torch.full((B,), t, device=DEV)
        eps = d16(xs, tt, sem, torch.full((B,), i, device=DEV))
        xs, x0 = sch.get_ddim_step(xs, tt, torch.full((B,), max(t - 250, 0), device=DEV), eps)
    assert torch.equal(a, x0)


def test_bf16_sampler_vs_reference_autocast(golden):
    """The bar for the bf16 sampler comes from the reference itself: its own generate_mel under torch.autocast("cpu", bfloat16)
    against its fp32 run on the same weights, tokens and start noise (tests/golden/make_golden_r3.py: bf16_sampler).  bf16
    rounding of eps is amplified 64171x at t=999 where x0 is not clamped, so the error of ANY bf16 run is a
    distribution with a heavy tail; ours must be no worse than the reference's own (cmp_bf16_error_distributions)."""
    g = golden("bf16_sampler")
    H, L, heads = (int(v) for v in g["cfg"].tolist())
    cfg = CFG(hidden=H, layers=L, heads=heads, device=DEV)
    dec = EdgeDiffusionDecoder(cfg, compute_dtype="bf16")
    dec.load_state_dict(synth_state_dict(cfg, 2))
    dec = dec.to(DEV).eval()
    infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), torch.nn.Identity(), dec)
    ours = infer.generate_mel(cu(g["sem_idx"]), 4, x_T=cu(g["x_T"])).cpu()
    ref = g["out_f32"].double()
    e_ours, e_ref = (ours.double() - ref).abs(), (g["out_autocast"].double() - ref).abs()
    cmp_bf16_error_distributions(e_ours, e_ref, "bf16 4-step sampler (hidden 64) vs the reference's fp32 run")
    assert float(ours.abs().max()) <= 3.0 and bool(torch.isfinite(ours).all())


def cmp_bf16_error_distributions(e_ours, e_ref, what):
    """ours vs the reference's own autocast(bf16) run, both as |x - x_fp32|.  A bf16 rounding of eps (~1e-2) is amplified 64171x at
    t = 999 where x0 is not clamped, so a few elements of ANY bf16 run flip across the whole clamp range: the
    maximum is the clamp range (5.9 of 6) for both runs whatever band is cut out -- round 3's "maximum outside a k = 64 band"
    compared 5.90 with 5.79 and could not fail, so it is gone.  What binds: the median, the 90th and the 99th percentile (each at
    most 1.1 x the reference's), and the NUMBER of flipped elements (error > 0.5: at most 1.25 x the reference's + 2)."""
    q = lambda e, p: float(e.flatten().quantile(p))
    big = lambda e: int((e > 0.5).sum())
    print(f"{what}: ours median {q(e_ours, .5):.2e} p90 {q(e_ours, .9):.2e} p99 {q(e_ours, .99):.2e} flipped {big(e_ours)} | reference "
          f"autocast(bf16): median {q(e_ref, .5):.2e} p90 {q(e_ref, .9):.2e} p99 {q(e_ref, .99):.2e} flipped {big(e_ref)} of {e_ref.numel()}")
    for p_ in (.5, .9, .99):
        assert q(e_ours, p_) <= 1.1 * q(e_ref, p_), (what, p_, q(e_ours, p_), q(e_ref, p_))
    assert big(e_ours) <= 1.25 * big(e_ref) + 2, (what, big(e_ours), big(e_ref))


def test_bf16_sampler_cfg3_shape_vs_reference_autocast(golden):
    """The same bar at BASELINE config 3's decoder shape (hidden 256, 8 layers, 8 heads; one utterance, T = 512), from the
    reference's own fp32 and autocast(bf16) runs of the 4-step sampler (tests/golden/make_golden_r4.py)."""
    g = golden("bf16_sampler_cfg3")
    H, L, heads = (int(v) for v in g["cfg"].tolist())
    cfg = CFG(hidden=H, layers=L, heads=heads, device=DEV)
    dec = EdgeDiffusionDecoder(cfg, compute_dtype="bf16")
    dec.load_state_dict(synth_state_dict(cfg, 1))
    dec = dec.to(DEV).eval()
    infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), torch.nn.Identity(), dec)
    ours = infer.generate_mel(cu(g["sem_idx"]), 4, x_T=cu(g["x_T"])).cpu()
    ref = g["out_f32"].double()
    cmp_bf16_error_distributions((ours.double() - ref).abs(), (g["out_autocast"].double() - ref).abs(),
                                 "bf16 4-step sampler, config-3 decoder shape, vs the reference's fp32 run")
    assert float(ours.abs().max()) <= 3.0 and bool(torch.isfinite(ours).all())


def _cfg3_bf16():
    cfg = CFG(hidden=256, layers=8, heads=8, device=DEV)
    dec = EdgeDiffusionDecoder(cfg, max_len=1024, compute_dtype="bf16")
    dec.load_state_dict(synth_state_dict(cfg, 1, max_pos=1024))
    return cfg, dec.to(DEV).eval()


def test_bf16_full_size_config3_properties():
    """BASELINE config 3 at FULL size (hidden 256, L = 8, heads 8, B = 256, T = 1024, bf16: 2 048 blocks per launch, q / k / v^T
    ping-pong, the block-shared LDS-DMA weight ring, the k_ctx16 image stores): too big for the oracle -> size-independent
    properties, as for config 2: run twice bitwise equal; slices {0, 131, 255} of the big batch == the same utterances in a small
    batch (bitwise); finite and clamped."""
    cfg, dec = _cfg3_bf16()
    infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), torch.nn.Identity(), dec)
    gen = torch.Generator().manual_seed(33)
    B, S = 256, 512
    sem = torch.randint(0, 512, (B, S), generator=gen).to(DEV)
    x = torch.randn(B, 2 * S, 80, generator=gen).to(DEV)
    big = infer.generate_mel(sem, 4, x_T=x)
    assert big.shape == (B, 1024, 80) and bool(torch.isfinite(big).all()) and float(big.abs().max()) <= 3.0
    assert torch.equal(big, infer.generate_mel(sem, 4, x_T=x))
    idx = [0, 131, 255]
    small = infer.generate_mel(sem[idx].contiguous(), 4, x_T=x[idx].contiguous())
    assert torch.equal(small, big[idx])
    # one utterance of the big batch, single forward, against the CPU ORACLE (fp32; one utterance of T = 1024 is ~20 GFLOP: seconds)
    sd = synth_state_dict(cfg, 1, max_pos=1024)
    t, si = torch.full((1,), 600), torch.full((1,), 1)
    e16 = dec(x[131:132].contiguous(), cu(t), sem[131:132].contiguous(), cu(si)).cpu()
    ref = O.decoder_forward(sd, x[131:132].cpu(), t, sem[131:132].cpu(), si, heads=8, window=cfg.attn_window_size)
    print(f"config 3, utterance 131 of the full batch, forward vs the oracle: rms {rms(e16, ref):.2e} max {max_abs(e16, ref):.2e}")
    assert rms(e16, ref) < BF16_RMS_TOL and max_abs(e16, ref) < BF16_MAX_TOL


def test_bf16_wide_instance(golden, tmp_path):
    """SCRATCH test since round 4: the 64-frame instance was measured at the default instance's speed and left the product library
    (-DEDTTS_EXPERIMENTS -DEDTTS16_WIDE_BUILD=1 builds it); runs only when EDTTS_TEST_WIDE_LIB names such a build.
    The 64-frames-per-wave bf16 instance (EDTTS16_WIDE=1; the library reads the switch once per process, hence the child): the
    config-3 shape against the reference's fp32 golden output and against the default instance on the same input; a 4-step
    sampler run whose utterances straddle blocks: run twice bitwise equal, probed utterances == the same utterances alone
    (bitwise), and bf16-level agreement with the default instance."""
    import os
    import subprocess
    import sys
    wide_lib = os.environ.get("EDTTS_TEST_WIDE_LIB")
    if not wide_lib:
        pytest.skip("the 64-frame bf16 instance is an experiment build (set EDTTS_TEST_WIDE_LIB to a -DEDTTS16_WIDE_BUILD=1 library)")
    out = str(tmp_path / "wide.npz")
    env = dict(os.environ, EDTTS16_WIDE="1", EDTTS_LIB=wide_lib)
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "bf16_wide_child.py")
    r = subprocess.run([sys.executable, child, out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    w = {k: torch.from_numpy(v) for k, v in np.load(out).items()}
    g = golden("forward_cfg3")
    cfg, dec = _cfg3_bf16()
    e_def = dec(cu(g["x_t"]), cu(g["t"]), cu(g["sem_idx"]), cu(g["step_idx"])).cpu()
    assert rms(w["eps"], g["eps"]) < BF16_RMS_TOL and max_abs(w["eps"], g["eps"]) < BF16_MAX_TOL
    assert rms(w["eps"], g["eps"]) <= rms(e_def, g["eps"]) * 1.05
    if os.environ.get("EDTTS16_WIDE") != "1":  # (unless this process runs the 64-frame instance itself)
        assert not torch.equal(w["eps"], e_def), "the child did not run another instance" 
    assert torch.equal(w["big"], w["again"]) and torch.equal(w["alone"], w["big"][[0, 11, 23]])
    for T2, S2, win, e_rms, e_max, finite in w["geo"].tolist():  # ragged geometries against the CPU oracle (see the child)
        print(f"wide instance, T={int(T2)} S={int(S2)} window={int(win)}: rms {e_rms:.2e} max {e_max:.2e}")
        assert finite == 1.0 and e_rms < BF16_RMS_TOL and e_max < BF16_MAX_TOL, (T2, S2, win, e_rms, e_max)
    infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), torch.nn.Identity(), dec)
    gen = torch.Generator().manual_seed(22)
    sem = torch.randint(0, 512, (24, 384), generator=gen).to(DEV)
    x = torch.randn(24, 768, 80, generator=gen).to(DEV)
    ref = infer.generate_mel(sem, 4, x_T=x).cpu()
    d = (w["big"] - ref).abs()
    print(f"wide vs default instance, 4-step sampler: rms {rms(w['big'], ref):.2e} median {float(d.median()):.2e} max {float(d.max()):.2e}")
    # (two bf16 runs differ like either differs from fp32: a heavy tail from the t=999 amplification, SURVEY.md F5.  The bar for the
    # median is what the reference's own autocast(bf16) run differs from its fp32 run at this decoder shape, doubled for two bf16 runs)
    gs = golden("bf16_sampler_cfg3")
    bar = 2.0 * float((gs["out_autocast"].double() - gs["out_f32"].double()).abs().median())
    assert bool(torch.isfinite(w["big"]).all()) and float(w["big"].abs().max()) <= 3.0 and float(d.median()) < bar, (float(d.median()), bar)


def test_bf16_no_cross_block_hazard_when_utterances_straddle_block_rounds():
    """The bf16 twin of test_no_cross_block_hazard...: T = 768 -> 6 blocks per utterance, B = 64 -> 384 blocks; blocks of one launch
    run at different times and a block's QKV tail writes the NEXT layer's K / V^T images while later neighbours still read this
    layer's halo tiles -- hence the two image sets.  Every probed utterance of the batch must equal the same utterance run alone."""
    cfg, dec = _cfg3_bf16()
    gen = torch.Generator().manual_seed(22)
    B, S = 64, 384
    x = torch.randn(B, 2 * S, 80, generator=gen).to(DEV)
    sem = torch.randint(0, 512, (B, S), generator=gen).to(DEV)
    t = torch.full((B,), 600, device=DEV)
    si = torch.full((B,), 1, device=DEV)
    big = dec(x, t, sem, si)
    assert bool(torch.isfinite(big).all())
    for u in (0, 5, 13, 21, 29, 37, 45, 53, 61, 63):
        solo = dec(x[u:u + 1].contiguous(), t[:1], sem[u:u + 1].contiguous(), si[:1])
        assert torch.equal(solo[0], big[u]), u


def test_ffn_mult_1_and_4(golden):
    """config.py:99 ffn_mult (FFN hidden width ffn_mult * H; the reference default 2 is what every other test runs): 4 and 1 against
    reference goldens on the fp32 path, and ffn_mult = 4 through the bf16 kernels against the oracle."""
    g = golden("forward_ffn_mult")
    for tag, mult in (("m4", 4), ("m1", 1)):
        cfg = CFG(hidden=32, heads=2, layers=2, ffn_mult=mult, device=DEV)
        dec = make_decoder(cfg, 8)
        e = dec(cu(g[f"{tag}_x_t"]), cu(g[f"{tag}_t"]), cu(g[f"{tag}_sem_idx"]), cu(g[f"{tag}_step_idx"])).cpu()
        assert max_abs(e, g[f"{tag}_eps"]) < FWD_TOL, (tag, max_abs(e, g[f"{tag}_eps"]))
    cfg = CFG(hidden=64, heads=2, layers=2, ffn_mult=4, device=DEV)
    sd = synth_state_dict(cfg, 7)
    gen = torch.Generator().manual_seed(12)
    x, sem = torch.randn(2, 80, 80, generator=gen), torch.randint(0, 512, (2, 40), generator=gen)
    t, si = torch.tensor([700, 30]), torch.tensor([1, 3])
    ref = O.decoder_forward(sd, x, t, sem, si, heads=2)
    dec = EdgeDiffusionDecoder(cfg, compute_dtype="bf16")
    dec.load_state_dict(sd)
    e = dec.to(DEV).eval()(cu(x), cu(t), cu(sem), cu(si)).cpu()
    assert rms(e, ref) < BF16_RMS_TOL and max_abs(e, ref) < BF16_MAX_TOL, (rms(e, ref), max_abs(e, ref))
    with pytest.raises(Exception, match="ffn_mult"):
        bad = CFG(hidden=32, heads=2, layers=1, ffn_mult=5, device=DEV)
        make_decoder(bad, 8)(torch.zeros(1, 32, 80, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV), torch.zeros(1, 16, dtype=torch.long, device=DEV))


def test_bf16_unsupported_head_dim_raises():
    from edge_diffusion_tts_amd.native import EdttsError
    cfg = CFG(device=DEV)  # head_dim 40
    dec = EdgeDiffusionDecoder(cfg, compute_dtype="bf16")
    dec.load_state_dict(synth_state_dict(cfg, 0))
    dec = dec.to(DEV).eval()
    with pytest.raises(EdttsError, match="head_dim 32"):
        dec(torch.zeros(1, 32, 80, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV), torch.zeros(1, 16, dtype=torch.long, device=DEV))
    with pytest.raises(ValueError):
        EdgeDiffusionDecoder(cfg, compute_dtype="fp8")


def test_inpaint_samplers(golden):
    """the long-form pipeline's two samplers (inference_pipeline.py:97-196) through edtts_sample_inpaint,
    against the reference's own closures run on CPU with the same noise draws: student (4 steps, step index 3) with and without a
    known tail, teacher refine with in-painting, with classifier-free guidance (two decoder passes per step) and without a tail."""
    from edge_diffusion_tts_amd import InpaintSampler
    g = golden("inpaint")
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    smp = InpaintSampler(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), dec)
    ov = int(g["overlap_len"])
    feats, known = cu(g["sem_features"]), cu(g["known_mel"])
    out = smp.inpaint_student_sample(tuple(g["stu_known_x_init"].shape), feats, known, ov, 4, x_init=cu(g["stu_known_x_init"]),
                                     noise_k=cu(g["stu_known_noise_k"])).cpu()
    assert max_abs(out, g["stu_known_out"]) < 5e-4 and torch.equal(out[:, :ov], g["known_mel"])
    out = smp.inpaint_student_sample(tuple(g["stu_free_x_init"].shape), feats, None, 0, 3, x_init=cu(g["stu_free_x_init"])).cpu()
    assert max_abs(out, g["stu_free_out"]) < 5e-4
    worst = 0.0
    for tag, kn in (("tea_known", True), ("tea_cfg", True), ("tea_free_cfg", False)):
        n, strength, scale = g[f"{tag}_params"].tolist()
        out = smp.inpaint_teacher_refine(cu(g["x_coarse"]), feats, known if kn else None, ov if kn else 0, strength, int(n), scale,
                                         noise=cu(g[f"{tag}_noise"]), noise_k=cu(g[f"{tag}_noise_k"]) if kn else None).cpu()
        err = max_abs(out, g[f"{tag}_out"])
        worst = max(worst, err)
        assert err < 5e-4, (tag, err)
        if kn:
            assert torch.equal(out[:, :ov], g["known_mel"])
    print(f"in-painting samplers vs the reference's closures: worst max-abs {worst:.2e}")
    # library noise path (no injected draws): deterministic per seed, batch of 2 chunks, finite, tail forced
    f2, k2 = feats.repeat(2, 1, 1), known.repeat(2, 1, 1)
    a = smp.inpaint_teacher_refine(cu(g["x_coarse"]).repeat(2, 1, 1), f2, k2, ov, 0.5, 4, 1.5, seed=3)
    b = smp.inpaint_teacher_refine(cu(g["x_coarse"]).repeat(2, 1, 1), f2, k2, ov, 0.5, 4, 1.5, seed=3)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all()) and torch.equal(a[:, :ov], k2)
    with pytest.raises(IndexError):
        smp.inpaint_teacher_refine(cu(g["x_coarse"]), feats, None, 0, 1.0, 4)   # t_start = 1000: outside the tables, as in the reference


@pytest.mark.gpu
def test_longform_stitch(golden):
    """inference_pipeline.py:296-367 through InpaintSampler.generate_long: three chunks (the last one ragged) refined with
    classifier-free guidance, each conditioned on the previous tail, de-normalised with per-chunk statistics, exp, cross-faded in
    the LINEAR domain and divided by the summed window -- against the reference's own loop statement run on its own closures with
    the same draws (tests/golden/make_golden_r3.py).  Tolerance 5e-4 of the largest linear-mel value."""
    from edge_diffusion_tts_amd import InpaintSampler
    g = golden("longform_stitch")
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    smp = InpaintSampler(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), dec)
    n_chunks = g["latent_slices"].shape[0]
    draws = [{k: cu(g[f"c{c}_{k}"]) for k in ("x_coarse", "noise", "noise_k") if f"c{c}_{k}" in g} for c in range(n_chunks)]
    stats = [(cu(g[f"c{c}_mean"]), cu(g[f"c{c}_std"])) for c in range(n_chunks)]
    steps, strength, scale = g["params"].tolist()
    total, chunk, ov, hop_len, sr = (int(v) for v in g["geometry"].tolist())
    out = smp.generate_long(cu(g["z_q_global"]), total, chunk, ov, stats, strength=strength, steps=int(steps), cfg_scale=scale,
                            hop_length=hop_len, sample_rate=sr, draws=draws).cpu()
    ref = g["final_mel"]
    assert out.shape == ref.shape == (cfg.n_mels, total)
    err = max_abs(out, ref)
    print(f"long-form stitch vs the reference's loop: max-abs {err:.2e} of a {float(ref.abs().max()):.3f} peak")
    assert err <= 5e-4 * float(ref.abs().max())
    # library noise path: deterministic per seed, finite, non-negative (a weighted mean of exponentials); no overlap = plain concatenation
    a = smp.generate_long(cu(g["z_q_global"]), total, chunk, ov, stats, strength=strength, steps=2, hop_length=hop_len, sample_rate=sr, seed=4)
    b = smp.generate_long(cu(g["z_q_global"]), total, chunk, ov, stats, strength=strength, steps=2, hop_length=hop_len, sample_rate=sr, seed=4)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all()) and float(a.min()) >= 0.0
    flat = smp.generate_long(cu(g["z_q_global"]), 96, 48, 0, stats[:2], strength=strength, steps=2, hop_length=hop_len, sample_rate=sr, seed=4)
    assert flat.shape == (cfg.n_mels, 96) and bool(torch.isfinite(flat).all())
    with pytest.raises(ValueError):
        smp.generate_long(cu(g["z_q_global"]), total, chunk, chunk, stats)


# ---------------------------------------------------------------------------------------------------------------
# mel post-processing -- PARITY UNPINNED: torchaudio (what the reference calls) cannot be installed
# offline, so the HIP kernels are checked against the oracle's restatement of torchaudio's published algorithm.
# ---------------------------------------------------------------------------------------------------------------
def _post_setup(B, T, seed):
    cfg = CFG(device=DEV)
    g = torch.Generator().manual_seed(seed)
    mel_n = torch.randn(B, T, cfg.n_mels, generator=g).clamp(-3, 3)
    mean = -5.0 + 0.5 * torch.randn(B, 1, cfg.n_mels, generator=g)
    std = 1.5 + 0.2 * torch.rand(B, 1, cfg.n_mels, generator=g)
    fb = O.melscale_fbanks(cfg.n_fft // 2 + 1, cfg.f_min, cfg.f_max, cfg.n_mels, cfg.sample_rate)
    return cfg, g, mel_n, mean, std, fb


def test_inverse_mel_scale_vs_oracle():
    from edge_diffusion_tts_amd import InverseMelScale
    cfg, g, mel_n, mean, std, fb = _post_setup(3, 37, 1)
    inv = InverseMelScale(n_stft=513, n_mels=80, sample_rate=16000, f_min=0.0, f_max=8000.0).to(DEV)
    assert torch.equal(inv.fb.cpu(), fb)
    lin_mel = torch.exp(mel_n * std + mean).transpose(1, 2)
    ref = O.inverse_mel_scale(lin_mel, fb)
    got = inv(cu(lin_mel.contiguous())).cpu()
    scale = float(ref.abs().max())
    assert got.shape == ref.shape == (3, 513, 37) and max_abs(got, ref) < 2e-5 * scale + 1e-9
    fused = inv.from_normalized(cu(mel_n), cu(mean), cu(std)).cpu()
    assert max_abs(fused, ref) < 2e-5 * scale + 1e-9 and float(fused.min()) >= 0.0


def test_griffin_lim_vs_oracle():
    """torchaudio.functional.griffinlim restated: same initial angles -> same waveform.  A few iterations agree to rounding; 32
    iterations (the reference's setting) of the non-linear phase projection amplify fp32 FFT rounding, so the bar there is the
    relative L2 error."""
    from edge_diffusion_tts_amd import GriffinLim
    cfg, g, mel_n, mean, std, fb = _post_setup(2, 40, 2)
    spec = O.inverse_mel_scale(torch.exp(mel_n * std + mean).transpose(1, 2), fb)
    a0 = torch.complex(torch.rand(spec.shape, generator=g), torch.rand(spec.shape, generator=g))
    for n_iter, tol in ((0, 1e-5), (2, 1e-4), (32, 1e-3)):
        ref = O.griffin_lim(spec, cfg.n_fft, cfg.hop_length, cfg.win_length, n_iter, angles0=a0)
        gl = GriffinLim(n_fft=cfg.n_fft, n_iter=n_iter, win_length=cfg.win_length, hop_length=cfg.hop_length, power=2.0).to(DEV)
        got = gl(cu(spec), angles0=cu(a0)).cpu()
        rel = float((got - ref).norm() / ref.norm())
        print(f"griffin-lim n_iter={n_iter}: relative L2 error {rel:.2e} (max-abs {max_abs(got, ref):.2e}, signal max {float(ref.abs().max()):.2e})")
        assert got.shape == ref.shape == (2, cfg.hop_length * 39) and rel < tol, (n_iter, rel)
    # library-drawn initial phases: deterministic per seed, finite, spectrally consistent (re-analysis magnitude close to the target)
    gl = GriffinLim(n_fft=cfg.n_fft, n_iter=32, win_length=cfg.win_length, hop_length=cfg.hop_length).to(DEV)
    w1, w2 = gl(cu(spec), seed=5), gl(cu(spec), seed=5)
    assert torch.equal(w1, w2) and bool(torch.isfinite(w1).all()) and not torch.equal(w1, gl(cu(spec), seed=6))


def test_mel_vocoder_end_to_end():
    """generate_sample.py:115-145 as one object, on the sampler's own output: generate_mel -> denormalise -> exp -> inverse mel
    -> Griffin-Lim, against the oracle pipeline on the same mel and the same initial angles."""
    from edge_diffusion_tts_amd import MelVocoder
    cfg, g, _, mean, std, fb = _post_setup(2, 64, 3)
    dec = make_decoder(cfg, 0)
    infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), torch.nn.Identity(), dec)
    sem = torch.randint(0, 512, (2, 32), generator=g)
    mel_n = infer.generate_mel(cu(sem), 4, seed=9)
    voc = MelVocoder(cfg, n_iter=4).to(DEV)
    a0 = torch.complex(torch.rand(2, 513, 64, generator=g), torch.rand(2, 513, 64, generator=g))
    wav = voc(mel_n, cu(mean), cu(std), angles0=cu(a0)).cpu()
    spec_ref, wav_ref = O.mel_to_waveform(mel_n.cpu(), mean, std, fb, cfg.n_fft, cfg.hop_length, cfg.win_length, 4, angles0=a0)
    rel = float((wav - wav_ref).norm() / wav_ref.norm())
    assert wav.shape == (2, cfg.hop_length * 63) and rel < 1e-3, rel


def test_small_batch_instance_equals_the_large_one():
    """Small grids (fewer than one 128-frame block per two CUs) run the 16-frames-per-wave instance: same arithmetic per frame, so
    an utterance gives bitwise the same mel whether it is sampled alone (small instance), in a batch of 32 (small instance,
    BASELINE config 2 strong-scaled over 8 GPUs) or in a batch of 80 (32-frame instance); also for windows off the tile grid and
    ragged lengths, and through the DDPM and multistep tails."""
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    sch = DiffusionSchedule(cfg.diff_steps).to(DEV)
    infer = EdgeInference(cfg, sch, torch.nn.Identity(), dec)
    gen = torch.Generator().manual_seed(40)
    B, S = 80, 256
    sem = torch.randint(0, 512, (B, S), generator=gen).to(DEV)
    x = torch.randn(B, 2 * S, 80, generator=gen).to(DEV)
    big = infer.generate_mel(sem, 4, x_T=x)                                   # 320 blocks: 32-frame instance
    mid = infer.generate_mel(sem[:32].contiguous(), 4, x_T=x[:32].contiguous())   # 128 blocks -> 16-frame instance
    one = infer.generate_mel(sem[7:8].contiguous(), 4, x_T=x[7:8].contiguous())
    assert torch.equal(mid, big[:32]) and torch.equal(one[0], big[7])
    for window, T in ((37, 150), (None, 96), (5, 70)):
        cfgw = CFG(device=DEV, attn_window_size=window)
        decw = make_decoder(cfgw, 2)
        Bw = 1100 // T * 8  # enough tiles for the 32-frame instance
        xs = torch.randn(Bw, T, 80, generator=gen).to(DEV)
        ss = torch.randint(0, 512, (Bw, T // 2), generator=gen).to(DEV)
        tt = torch.randint(0, 1000, (Bw,), generator=gen).to(DEV)
        e_big = decw(xs, tt, ss, None)
        e_small = decw(xs[:2].contiguous(), tt[:2].contiguous(), ss[:2].contiguous(), None)
        assert torch.equal(e_small, e_big[:2]), (window, T)


def test_substreams_do_not_change_results():
    """A batch of at least two rounds of waves is cut into two halves that run on two streams (include/edtts.h:
    edtts_set_substreams).  Every utterance is computed alone, so the cut must be invisible: bitwise equal results with the cut
    off and on, eagerly and from a captured hipGraph (the fork / join events are part of the capture), for an odd batch (halves
    of different size), for the DDPM sampler's Philox noise (keyed by the GLOBAL utterance index), and an out-of-range token in
    the SECOND half is still recorded in word 0 of the caller's workspace."""
    from edge_diffusion_tts_amd import native
    cfg = CFG(device=DEV)
    dec = make_decoder(cfg, 0)
    infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), torch.nn.Identity(), dec)
    gen = torch.Generator().manual_seed(12)
    B, S = 129, 256   # 129 * 16 = 2064 waves >= 2 * 1024 SIMDs: the cut applies; halves of 65 and 64 utterances
    sem = torch.randint(0, 512, (B, S), generator=gen).to(DEV)
    x = torch.randn(B, 2 * S, 80, generator=gen).to(DEV)
    prev = native.set_substreams(1)
    try:
        one = infer.generate_mel(sem, 4, x_T=x)
        one_d = infer.sample_ddpm(sem, 3, x_T=x, seed=11)
        native.set_substreams(2)
        two = infer.generate_mel(sem, 4, x_T=x)
        two_d = infer.sample_ddpm(sem, 3, x_T=x, seed=11)
        assert torch.equal(one, two) and torch.equal(one_d, two_d)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out_g = infer.generate_mel(sem, 4, x_T=x)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out_g, one)
        # the caller's stream sees ordinary stream semantics: work enqueued right behind the call reads finished results
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            two_s = infer.generate_mel(sem, 4, x_T=x)
            chk = two_s.clone()
        s.synchronize()
        assert torch.equal(chk, one)
        bad = sem.clone()
        bad[B - 1, 5] = 512
        infer.generate_mel(bad, 4, x_T=x)
        ws = dec.workspace(B, 2 * S, S, 4, x.device)
        assert native.index_errors(ws) == 1 and native.index_errors(ws) == 0
        # three sub-batches (385 utterances = 6160 waves: two rounds each allow three; 129 + 128 + 128) against one piece
        native.set_substreams(4)
        B3 = 385
        assert native.substreams_for(dec.dims(), B3, 2 * S) == 3
        sem3 = torch.randint(0, 512, (B3, S), generator=gen).to(DEV)
        three = infer.generate_mel(sem3, 2, seed=21)
        native.set_substreams(1)
        assert torch.equal(three, infer.generate_mel(sem3, 2, seed=21))
    finally:
        native.set_substreams(prev)


@pytest.mark.parametrize("B,S,window,hidden,ffn_mult", [(1, 128, 64, 160, 2), (3, 37, 64, 160, 2), (2, 256, 64, 160, 2), (1, 48, 37, 160, 2),
                                                         (5, 16, None, 160, 2), (2, 45, 64, 256, 2), (2, 40, 64, 160, 1), (1, 70, 64, 160, 3),
                                                         (2, 33, 64, 160, 4)])
def test_cooperative_instances_equal_the_one_wave_ones(B, S, window, hidden, ffn_mult):
    """Small grids run the transformer layers with 2 or 4 waves per frame tile (csrc/edtts_coop.h; chosen from the tile count,
    include/edtts.h: edtts_set_coop).  The split is by heads and by OUTPUT tiles only, so no sum changes: every forced instance
    (16-frame tiles x 4 waves, 32-frame tiles x 4 and x 2 waves -- the latter with an odd tile count: a block's surplus tile) must
    equal the one-wave kernels bitwise, for the eps tail (decoder forward), the fused DDIM tail (generate_mel), the DDPM tail with
    in-kernel noise and the multistep-solver tail; and the automatic choice must be one of them."""
    from edge_diffusion_tts_amd import DPMSolverPP, native
    # (ffn_mult 1 / 3 / 4: the FFN's hidden tiles pass through LDS in groups of 2 * hidden / 16 -- a short only group, a full group + a
    # short one, two full groups)
    cfg = (CFG(device=DEV, attn_window_size=window, ffn_mult=ffn_mult) if hidden == 160 else
           CFG(device=DEV, attn_window_size=window, hidden=256, heads=8, layers=2))
    dec = make_decoder(cfg, 3)  # (hidden 256: 16- and 32-frame tiles x 4 waves; 32-frame x 2 does not fit its LDS twice per CU and falls back)
    sch = DiffusionSchedule(cfg.diff_steps).to(DEV)
    infer = EdgeInference(cfg, sch, torch.nn.Identity(), dec)
    gen = torch.Generator().manual_seed(100 + B + S)
    sem = torch.randint(0, 512, (B, S), generator=gen).to(DEV)
    x = torch.randn(B, 2 * S, 80, generator=gen).to(DEV)
    t = torch.randint(0, 1000, (B,), generator=gen).to(DEV)
    si = torch.randint(0, 16, (B,), generator=gen).to(DEV)
    feats = torch.randn(B, S, cfg.semantic_dim, generator=gen).to(DEV)

    def run():
        return (dec(x, t, sem, si), infer.generate_mel(sem, 4, x_T=x), infer.sample_ddpm(sem, 3, x_T=x, seed=5),
                DPMSolverPP(sch, order=3).sample(dec, x, feats, num_steps=4))

    prev = native.set_coop(0)
    try:
        ref = run()
        for mode in (14, 24, 22, -1):
            native.set_coop(mode)
            out = run()
            for name, a, b_ in zip(("forward", "generate_mel", "sample_ddpm", "dpm_solver"), ref, out):
                assert torch.equal(a, b_), f"cooperative instance {mode}: {name} differs from the one-wave kernels (max {max_abs(a.cpu(), b_.cpu()):.3e})"
    finally:
        native.set_coop(prev)


def test_bench_two_ranks_rehearsal():
    """The multi-rank path of bench.py with the REAL sampler, as the driver starts it (`python bench.py --gpus 2`, no launcher
    around it): a fresh child process spawns two ranks before anything touches the GPU; with EDTTS_BENCH_REHEARSAL=1 both share
    this box's one GPU and the all-gather runs over gloo.  Checks the contract of the line, not its numbers (not a scaling
    measurement): one JSON line, two ranks joined, an all-gather was timed, the weak-scaling aggregate is consistent."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EDTTS_BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "EDTTS_BENCH_STUB"):
        env.pop(k, None)
    # (128 utterances per rank: large enough for the two-stream cut of the sampler call, so the all-gather follows a forked / joined call)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "128",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["allgather_ms"] > 0 and d["allgather_bytes"] == 2 * 128 * 512 * 80 * 4
    assert d["config"]["batch_per_gpu"] == 128 and d["config"]["substreams"] == 2 and "REHEARSAL" in d["note"]
    assert abs(d["value"] - 2 * 128 * 512 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


def test_build_time_instance_192_6_80_vs_oracle():
    """A decoder shape that is not built in (hidden 192, 6 heads of 32, n_mels 80) arrives through the build-time instance list
    (__graft_entry__.DEFAULT_INSTANCES / EDTTS_INSTANCES): single forward and the 4-step sampler against the CPU oracle, and a shape
    that is in neither list is refused with the list in the message."""
    import os
    import __graft_entry__ as G
    if "192x6x80" not in os.environ.get("EDTTS_INSTANCES", G.DEFAULT_INSTANCES):
        pytest.skip("this library was built without the 192x6x80 instance")
    cfg = CFG(hidden=192, heads=6, layers=3, device=DEV)
    sd = synth_state_dict(cfg, 4)
    dec = EdgeDiffusionDecoder(cfg)
    dec.load_state_dict(sd)
    dec = dec.to(DEV).eval()
    gen = torch.Generator().manual_seed(77)
    B, S = 3, 41   # T = 82: ragged against the 32-frame tiles
    sem = torch.randint(0, 512, (B, S), generator=gen)
    x = torch.randn(B, 2 * S, 80, generator=gen)
    t = torch.tensor([999, 500, 3])
    si = torch.tensor([0, 7, 15])
    e = dec(cu(x), cu(t), cu(sem), cu(si)).cpu()
    ref = O.decoder_forward(sd, x, t, sem, si, heads=6)
    assert max_abs(e, ref) < FWD_TOL, max_abs(e, ref)
    infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), torch.nn.Identity(), dec)
    out = infer.generate_mel(cu(sem), 4, x_T=cu(x)).cpu()
    tr = []
    ref_out = O.generate_mel(sd, O.schedule_tables(1000)["alpha_bar"], sem, x, 4, heads=6, trace=tr)
    e2e_check(out, ref_out, amplification_band(x, tr[0]["eps"]), "192/6/80 instance, 4-step sampler vs oracle")
    from edge_diffusion_tts_amd import native
    bad = EdgeDiffusionDecoder(CFG(hidden=224, heads=7, device=DEV)).to(DEV).eval()
    with pytest.raises(native.EdttsError, match="192/6/80"):
        bad(torch.zeros(1, 32, 80, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV), torch.zeros(1, 16, dtype=torch.long, device=DEV))


def test_build_time_bf16_instance_128_4_80_vs_oracle():
    """A bf16 decoder shape from the build-time list (EDTTS_INSTANCES_BF16 / __graft_entry__.DEFAULT_INSTANCES_BF16; head_dim 32
    shapes only): forward against the fp32 CPU oracle at the bf16 tolerances, sampler properties, and a head_dim-40 bf16 request is
    refused with a message that says why."""
    import os
    import __graft_entry__ as G
    from edge_diffusion_tts_amd import native
    if "128x4x80" not in os.environ.get("EDTTS_INSTANCES_BF16", G.DEFAULT_INSTANCES_BF16):
        pytest.skip("this library was built without the bf16 128x4x80 instance")
    cfg = CFG(hidden=128, heads=4, layers=3, device=DEV)
    sd = synth_state_dict(cfg, 6)
    dec = EdgeDiffusionDecoder(cfg, compute_dtype="bf16")
    dec.load_state_dict(sd)
    dec = dec.to(DEV).eval()
    gen = torch.Generator().manual_seed(78)
    B, S = 3, 53
    sem = torch.randint(0, 512, (B, S), generator=gen)
    x = torch.randn(B, 2 * S, 80, generator=gen)
    t = torch.tensor([999, 420, 7])
    si = torch.tensor([0, 5, 15])
    e = dec(cu(x), cu(t), cu(sem), cu(si)).cpu()
    ref = O.decoder_forward(sd, x, t, sem, si, heads=4)
    print(f"bf16 128/4/80 instance, forward vs the fp32 oracle: rms {rms(e, ref):.2e} max {max_abs(e, ref):.2e}")
    assert bool(torch.isfinite(e).all()) and rms(e, ref) < BF16_RMS_TOL and max_abs(e, ref) < BF16_MAX_TOL
    infer = EdgeInference(cfg, DiffusionSchedule(cfg.diff_steps).to(DEV), torch.nn.Identity(), dec)
    a = infer.generate_mel(cu(sem), 4, x_T=cu(x))
    assert torch.equal(a, infer.generate_mel(cu(sem), 4, x_T=cu(x))) and float(a.abs().max()) <= 3.0
    assert torch.equal(infer.generate_mel(cu(sem[1:2]).contiguous(), 4, x_T=cu(x[1:2]).contiguous())[0], a[1])
    with pytest.raises(native.EdttsError, match="head_dim 32"):
        bad = EdgeDiffusionDecoder(CFG(device=DEV), compute_dtype="bf16").to(DEV).eval()
        bad(torch.zeros(1, 32, 80, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV), torch.zeros(1, 16, dtype=torch.long, device=DEV))
