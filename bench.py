#!/usr/bin/env python3
"""bench.py -- headline benchmark of the DDIM sampler path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE EdgeInference.generate_mel call (4-step DDIM, CFG() decoder: hidden=160, L=4, heads=4, n_mels=80)
over one synthetic batch of B=256 utterances x T=512 mel frames per GPU (BASELINE.json configs[1]; with N GPUs the
job is configs[3]'s weak-scaling layout: 256 utterances per rank, the final mel batch all-gathered over RCCL).
Inputs (sem_idx, start noise) are resident in HBM before the timed region; weights are synthetic (no checkpoints
offline).  Rank 0 prints ONE JSON line.  value = whole-job mel-frames/s = N*B*T / (max-over-ranks time per step).

Extra legs (rank 0, N = 1 only):
  roofline      dominant kernel = the fused transformer-layer kernel (k_layer); achieved = algorithmic FLOPs per launch /
                its average launch duration measured live with HIP events on the launch stream (edtts_profile_*).
  cpu_baseline  the CPU oracle (oracle/edtts_oracle.py, a PyTorch-CPU port of the reference path -- the reference's
                Python cannot travel to the GPU box) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "edge-diffusion-tts_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

METRIC = "mel-frames/sec (whole node), 4-step DDIM, hidden=160 L=4, B=256 T=512"
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_* dense peak (no xf32 on gfx950)


def layer_flops_per_frame(H, M, S, T, W, last):
    """Algorithmic FLOPs (2*MAC, contractions only) one k_layer launch spends per mel frame (DESIGN.md "Work model"):
    proj + q_proj + out_proj + FFN (= 18 H^2 ... ) + banded self-attention + cross-attention + the tail
    (QKV of the next layer: 6 H^2, or final out_proj: 2 M H)."""
    nbar = sum(min(i + W, T - 1) - max(i - W, 0) + 1 for i in range(T)) / T
    body = 2 * H * H * (1 + 1 + 1) + 2 * H * (4 * H) + 2 * (2 * H) * H + 4 * nbar * H + 4 * S * H
    tail = 2 * M * H if last else 2 * H * (3 * H)
    return body + tail


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=512, help="mel frames per utterance (T = 2*S)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=128)  # ~10-15 s of CPU work on 16 threads
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    import __graft_entry__
    import torch.distributed as dist
    # EDTTS_BENCH_REHEARSAL=1: all ranks share cuda:0 and the collective runs over gloo -- exercises the multi-rank code path on
    # a one-GPU box; the numbers it prints are NOT a scaling measurement.
    rehearsal = os.environ.get("EDTTS_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if rank == 0:
        __graft_entry__.build()  # (re)build the HIP library once; the other ranks load it after the barrier
    if world > 1:
        dist.barrier()
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    from edge_diffusion_tts_amd import CFG, DiffusionSchedule, EdgeDiffusionDecoder, EdgeInference, native, synth_state_dict
    from edge_diffusion_tts_amd.parallel import gather_batch, generate_overlapped

    cfg = CFG(device="cuda")
    dec = EdgeDiffusionDecoder(cfg)
    sd = synth_state_dict(cfg, 0)
    dec.load_state_dict(sd)
    dec = dec.to(dev).eval()
    sch = DiffusionSchedule(cfg.diff_steps).to(dev)
    infer = EdgeInference(cfg, sch, torch.nn.Identity(), dec)

    B, T = args.batch, args.frames
    S = T // 2
    gen = torch.Generator().manual_seed(2 + rank)
    sem = torch.randint(0, cfg.codebook_size, (B, S), generator=gen).to(dev)
    x_T = torch.randn(B, T, cfg.n_mels, generator=torch.Generator().manual_seed(123 + rank)).to(dev)
    stream = torch.cuda.current_stream(dev)

    # EDTTS_BENCH_MICRO=M (>1, opt-in): run the local batch in M slices and overlap each slice's all-gather with the next
    # slice's compute (parallel.generate_overlapped).  Default 1: one all-gather after the whole local batch.
    micro = int(os.environ.get("EDTTS_BENCH_MICRO", "1"))

    def step():
        if world > 1 and micro > 1:
            return generate_overlapped(lambda s_, n_, x_: infer.generate_mel(s_, n_, x_T=x_), sem, x_T, 4, world * B, micro)
        mel = infer.generate_mel(sem, 4, x_T=x_T)
        if world > 1:
            mel = gather_batch(mel, world * B)  # the one collective of the path: final mel batch, RCCL over xGMI
        return mel

    for _ in range(args.warmup):
        step()
    n_layer_launches = args.steps * 4 * cfg.layers * 2  # a layer is one or two launches
    profile = world == 1 and os.environ.get("EDTTS_BENCH_NO_EVENTS", "0") != "1"
    if profile:
        native.profile_enable(n_layer_launches)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cpu" if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert out.shape == (world * B, T, cfg.n_mels) and bool(torch.isfinite(out[:2]).all())

    ms_per_step = dt / args.steps * 1e3
    frames_per_s = world * B * T / (dt / args.steps)
    result = {
        "metric": METRIC, "value": frames_per_s, "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"generate_mel 4-step DDIM, CFG() decoder hidden=160 L=4 heads=4 n_mels=80 window=64, "
                               f"B={B}/GPU T={T} S={S}, synthetic weights + tokens + noise",
                   "batch_per_gpu": B, "frames": T, "ddim_steps": 4,
                   "parallelism": (f"batch-sharded x{world}, all-gather of the final mel batch" + (f" overlapped with compute in {micro} slices" if micro > 1 else ""))
                   if world > 1 else "single GPU"},
        "mels_per_s": world * B / (dt / args.steps),
    }
    if rehearsal:
        result["note"] = "REHEARSAL: ranks share one GPU, gloo collective -- not a scaling measurement"

    if world == 1 and not profile:
        result["note"] = "EDTTS_BENCH_NO_EVENTS=1: roofline leg skipped"
    if profile:
        (ms0, n0), (ms1, n1) = native.profile_collect()
        native.profile_enable(0)
        frames = B * T
        H, M, W = cfg.hidden, cfg.n_mels, cfg.attn_window_size
        nbar = sum(min(i + W, T - 1) - max(i - W, 0) + 1 for i in range(T)) / T
        attn_flops = frames * (2 * H * H * 3 + 4 * nbar * H + 4 * S * H)              # proj, q_proj, out_proj + both attentions
        ffn_flops = [frames * (2 * H * 4 * H + 2 * 2 * H * H + (2 * M * H if l == cfg.layers - 1 else 6 * H * H)) for l in range(cfg.layers)]
        flops = cfg.layers * attn_flops + sum(ffn_flops)                               # one decoder forward, all layer kernels
        layer_ms = ms0 + ms1
        traffic = None
        pmc = os.path.join(REPO, "profiles", "r01_pmc_k_layer.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None

        def roof(name, fl_per_launch, ms, n):
            avg = ms / max(n, 1)
            ach = fl_per_launch / (avg * 1e-3) / 1e12
            return {"bound": "mfma", "kernel": name, "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_F32_MFMA_TFLOPS, "avg_launch_ms": avg, "launches_timed": n,
                    "algorithmic_gflop_per_launch": fl_per_launch / 1e9}
        if n1 == 0:   # fused layer kernel
            result["roofline"] = roof("k_layer (fused transformer layer)", flops / cfg.layers, ms0, n0)
        else:         # layer = attention half + FFN/tail half; the dominant one (by time) is the roofline kernel
            r_attn = roof("k_layer<PART_ATTN> (self + cross attention, projections)", attn_flops, ms0, n0)
            r_ffn = roof("k_layer<PART_FFN> (SwiGLU FFN + QKV / output tail)", sum(ffn_flops) / cfg.layers, ms1, n1)
            dom, oth = (r_attn, r_ffn) if ms0 >= ms1 else (r_ffn, r_attn)
            result["roofline"] = dom
            result["roofline_other_kernel"] = oth
            result["roofline_layer_pair"] = {"achieved": flops / cfg.layers / ((ms0 + ms1) / max(n0, 1) * 1e-3) / 1e12,
                                             "frac": flops / cfg.layers / ((ms0 + ms1) / max(n0, 1) * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                             "ms_per_layer": (ms0 + ms1) / max(n0, 1), "unit": "TFLOP/s"}
        result["roofline"]["traffic"] = traffic
        result["roofline"]["layer_kernels_share_of_step"] = layer_ms / args.steps / ms_per_step
        total_flops = 4 * flops + 4 * frames * 2 * cfg.n_mels * cfg.hidden + 4 * frames * 6 * cfg.hidden ** 2 \
            + cfg.layers * B * S * 3 * cfg.hidden ** 2
        result["whole_call"] = {"algorithmic_tflop": total_flops / 1e12, "tflops": total_flops / (dt / args.steps) / 1e12,
                                "frac_of_f32_mfma_peak": total_flops / (dt / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS}
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, sd, args.cpu_sample_batch, T)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def cpu_baseline(cfg, sd, Bc, T):
    """The CPU oracle (a PyTorch-CPU port of the reference path, validated against the reference's outputs in
    tests/test_oracle_vs_golden.py) timed on this host: one 4-step generate_mel on a bounded sample."""
    from oracle import edtts_oracle as O
    # this process's CPU share on the GPU box is 16 cores (torch would otherwise start one thread per host core)
    threads = int(os.environ.get("EDTTS_CPU_THREADS", str(min(16, os.cpu_count() or 1))))
    torch.set_num_threads(threads)
    S = T // 2
    g = torch.Generator().manual_seed(2)
    sem = torch.randint(0, cfg.codebook_size, (Bc, S), generator=g)
    x_T = torch.randn(Bc, T, cfg.n_mels, generator=g)
    ab = O.schedule_tables(cfg.diff_steps)["alpha_bar"]
    with torch.no_grad():
        O.generate_mel(sd, ab, sem[:2], x_T[:2], 4)  # warm-up
        t0 = time.perf_counter()
        O.generate_mel(sd, ab, sem, x_T, 4)
        dt = time.perf_counter() - t0
    return {"value": Bc * T / dt, "unit": "mel-frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle/edtts_oracle.generate_mel, 4-step DDIM, B={Bc} T={T} fp32, torch {torch.__version__} CPU, "
                      f"{threads} threads, 1 call after warm-up, {dt:.1f} s", "seconds": dt}


if __name__ == "__main__":
    main()
