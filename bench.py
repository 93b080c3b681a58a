#!/usr/bin/env python3
"""bench.py -- benchmark of the DDIM sampler path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C]

Headline (default, --config 2): a "step" is ONE EdgeInference.generate_mel call (4-step DDIM, CFG() decoder: hidden=160, L=4,
heads=4, n_mels=80) over one synthetic batch of B=256 utterances x T=512 mel frames per GPU (BASELINE.json configs[1]; with N
GPUs the job is configs[3]'s weak-scaling layout: 256 utterances per rank, the final mel batch all-gathered over RCCL).  The
whole call is inside the timed region, INCLUDING the start-noise draw (inference.py:33; here the library's Philox kernel);
sem_idx is resident in HBM, weights are synthetic (no checkpoints offline).  Rank 0 prints ONE JSON line;
value = whole-job mel-frames/s = N*B*T*K / (max-over-ranks wall time of the K steps).

Protocol: the library is built before anything touches the GPU; W untimed warm-up steps; EXACTLY K steps
bracketed by barrier + synchronize with all instrumentation off (value, ms_per_step = mean; the per-step HIP-event times of the
same loop give ms_per_step_median).  Extra legs, rank 0 at N = 1 only, all OUTSIDE the headline loop:
  roofline      a second loop with the library's per-launch HIP events on (edtts_profile_*): dominant kernel = the fused
                transformer-layer kernel k_layer; achieved = algorithmic FLOPs per launch / its average launch duration.
                traffic = HBM bytes per k_layer launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, each in its own
                child run of this script started BEFORE the parent initialises the GPU; gfx950 correction of
                MI355X_MICROARCH.md: FETCH_SIZE x2 for 16 B/lane streams) -- null when rocprofv3 is unavailable or --no-pmc.
  cpu_baseline  the CPU oracle (oracle/edtts_oracle.py, a PyTorch-CPU port of the reference path -- the reference's Python
                cannot travel to the GPU box) timed on this box's host cores on a bounded sample.

Other BASELINE.json configurations: --config 1 (B=1, T=256 latency), 3 (hidden=256 L=8 heads=8, B=256, T=1024; --dtype bf16|f32),
5 (1000-step DDPM sampler, B=64, T=512, one captured hipGraph per step).  Config 4 is `--gpus 8` of the default.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "edge-diffusion-tts_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

HEADLINE = "mel-frames/sec (whole node), 4-step DDIM, hidden=160 L=4, B=256 T=512"
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}  # /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks (no xf32 on gfx950)
PEAK_HBM_GBS = 8000.0

CONFIGS = {
    1: dict(hidden=160, layers=4, heads=4, B=1, T=256, sampler="ddim", num_steps=4, dtype="f32", steps=200),
    2: dict(hidden=160, layers=4, heads=4, B=256, T=512, sampler="ddim", num_steps=4, dtype="f32", steps=300),
    3: dict(hidden=256, layers=8, heads=8, B=256, T=1024, sampler="ddim", num_steps=4, dtype="bf16", steps=30),
    5: dict(hidden=160, layers=4, heads=4, B=64, T=512, sampler="ddpm", num_steps=1000, dtype="f32", steps=5),
}


def band_mean_keys(T, W):
    if W is None or W < 0:
        return float(T)
    return sum(min(i + W, T - 1) - max(i - W, 0) + 1 for i in range(T)) / T


def layer_flops_per_frame(H, M, S, T, W, last):
    """Algorithmic FLOPs (2*MAC, contractions only) one k_layer launch spends per mel frame (BASELINE.md "Work model"):
    proj + q_proj + out_proj (6 H^2) + FFN (8 H^2 + 4 H^2) + banded self-attention (4 nbar H) + cross-attention (4 S H) + the tail
    (QKV of the next layer: 6 H^2, or final out_proj: 2 M H)."""
    body = 18 * H * H + 4 * band_mean_keys(T, W) * H + 4 * S * H
    return body + (2 * M * H if last else 6 * H * H)


def call_flops(H, L, M, S, T, W, B, n_steps):
    """One sampler call: n_steps decoder forwards (in_proj + QKV(0) in the prologue, L layer kernels) + the context K/V once."""
    per_fwd = sum(layer_flops_per_frame(H, M, S, T, W, l == L - 1) for l in range(L)) + 2 * M * H + 6 * H * H
    return B * T * n_steps * per_fwd + L * B * S * 3 * H * H


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ------------------------------------------------------------------------------------------------ PMC traffic leg
def pmc_traffic(args):
    """HBM bytes per k_layer launch from two rocprofv3 PMC passes of a short child run of this script (separate passes: FETCH_SIZE
    and WRITE_SIZE do not fit one; no trace domains are combined with --pmc).  Must run before this process touches the GPU."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="edtts_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    res = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, counter)
            cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-child", "--config", str(args.config), "--dtype", args.dtype, "--steps", "2", "--warmup", "1"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {r.stderr.decode(errors='replace')[-200:]}"
            vals = [float(row["Counter_Value"]) for f in files for row in csv.DictReader(open(f))
                    if row["Counter_Name"] == counter and "k_layer" in row["Kernel_Name"]]
            if not vals:
                return None, f"no k_layer dispatch in the {counter} pass"
            res[counter] = sum(vals) / len(vals)
    except Exception as e:  # noqa: BLE001  (a profiler problem must not take the benchmark down)
        return None, f"PMC leg failed: {e!r}"
    finally:
        shutil.rmtree(out, ignore_errors=True)
    # FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports 1/2 of the bytes of a 16 B/lane streaming read on gfx950
    rd, wr = 2.0 * res["FETCH_SIZE"] * 1024, res["WRITE_SIZE"] * 1024
    return {"bytes": rd + wr, "read": rd, "write": wr}, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 2 --warmup 1`, mean over k_layer launches; FETCH_SIZE x2 (gfx950 16 B/lane correction)"


# ------------------------------------------------------------------------------------------------ self-launch
def spawn_ranks(n):
    """`python bench.py --gpus N` without an outer torchrun: start N copies of this script, one per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set as torch.distributed.run would), relay rank 0's JSON line, exit with the worst child status.
    The parent only compiles the library when it is stale (hipcc; __graft_entry__.compile_library imports neither torch nor the
    package and dlopens nothing) -- a process that has initialised the GPU must not fork workers.  A rank that dies takes the
    others down instead of leaving them waiting in a collective, and so does the parent's own death (interrupt, driver timeout):
    the exact PIDs started here are terminated, then killed."""
    import signal
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EDTTS_BENCH_SPAWNED="1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes needs it on this driver)
    import __graft_entry__
    __graft_entry__.compile_library()  # once, before any rank starts
    procs = []

    def on_signal(signum, _frame):
        raise KeyboardInterrupt(f"signal {signum}")

    old = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        for r in range(n):
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=None if r == 0 else subprocess.DEVNULL))
        live = list(procs)
        while live:
            time.sleep(0.2)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in live:  # exact PIDs we started
                        q.terminate()
    except KeyboardInterrupt:
        rc = rc or 130
    finally:
        for p in procs:  # whatever is still alive when we leave (normal exit: nothing)
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        for sig, h in old.items():
            signal.signal(sig, h)
    return rc


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default=None, choices=("f32", "bf16"))
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default: the config's)")
    ap.add_argument("--frames", type=int, default=None, help="mel frames per utterance (T = 2*S)")
    ap.add_argument("--substreams", type=int, default=None, choices=(1, 2, 3, 4, 5, 6, 7, 8),
                    help="most sub-batches (streams) a sampler call may be cut into (include/edtts.h: edtts_set_substreams); default: the library's (4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=128)  # ~10-15 s of CPU work on 16 threads
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    C = dict(CONFIGS[args.config])
    args.dtype = args.dtype or C["dtype"]
    B, T = args.batch or C["B"], args.frames or C["T"]
    S = T // 2
    steps = args.steps if args.steps is not None else C["steps"]
    warmup = args.warmup if args.warmup is not None else max(2, min(10, steps // 10))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))  # plain `python bench.py --gpus N`: this process only launches and relays
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    extras = world == 1 and not args.pmc_child
    # EDTTS_BENCH_STUB=1 (tests only): no GPU -- a stand-in sampler on the CPU and gloo, to exercise the launcher, the sharding,
    # the collective and the JSON contract of the multi-rank path where no GPU exists.  Its numbers mean nothing.
    stub = os.environ.get("EDTTS_BENCH_STUB", "0") == "1"

    # ---- everything that must precede the first GPU call: build, PMC child runs -------------------------------------------
    import __graft_entry__
    if rank == 0 and os.environ.get("EDTTS_BENCH_SPAWNED") != "1":
        __graft_entry__.build()  # (re)builds only when stale; the driver has run build() already
    traffic, traffic_src = None, "skipped"
    if extras and not stub and not args.no_pmc and not args.no_roofline:
        traffic, traffic_src = pmc_traffic(args)

    import torch
    import torch.distributed as dist
    # EDTTS_BENCH_REHEARSAL=1: all ranks share cuda:0 and the collective runs over gloo -- exercises the multi-rank code path on
    # a one-GPU box; the numbers it prints are NOT a scaling measurement.
    rehearsal = os.environ.get("EDTTS_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    rccl_ranks = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist.barrier()  # rank 0 has finished building
        # how many ranks the collective backend really joined: an all-reduce of ones over the group the gather will use
        one = torch.ones(1, dtype=torch.int64, device="cpu" if (stub or rehearsal) else torch.device("cuda", local_rank))
        dist.all_reduce(one)
        rccl_ranks = int(one.item())
        if rccl_ranks != world:
            raise SystemExit(f"collective backend joined {rccl_ranks} ranks, expected {world}")
    dev = torch.device("cpu") if stub else torch.device("cuda", local_rank if world > 1 else 0)
    if not stub:
        torch.cuda.set_device(dev)

    def sync():
        if not stub:
            torch.cuda.synchronize(dev)

    from edge_diffusion_tts_amd import CFG, DiffusionSchedule, EdgeDiffusionDecoder, EdgeInference, native, synth_state_dict
    from edge_diffusion_tts_amd.parallel import gather_batch, generate_overlapped

    cfg = CFG(device="cpu" if stub else "cuda", hidden=C["hidden"], layers=C["layers"], heads=C["heads"])
    max_len = max(1000, T)
    gen = torch.Generator().manual_seed(2 + rank)
    sem = torch.randint(0, cfg.codebook_size, (B, S), generator=gen).to(dev)
    micro = int(os.environ.get("EDTTS_BENCH_MICRO", "1"))  # >1 (opt-in): overlap each slice's all-gather with the next slice's compute
    gather_out = [None]
    counter = [0]
    gather_ms = []  # filled by the separate gather-timing loop below, never inside the headline loop
    if stub:
        sd = infer = None
        if C["sampler"] != "ddim":
            raise SystemExit("EDTTS_BENCH_STUB covers the DDIM configurations only")
    else:
        dec = EdgeDiffusionDecoder(cfg, max_len=max_len, **({"compute_dtype": args.dtype} if args.dtype != "f32" else {}))
        sd = synth_state_dict(cfg, 0, max_pos=max_len)
        dec.load_state_dict(sd)
        dec = dec.to(dev).eval()
        sch = DiffusionSchedule(cfg.diff_steps).to(dev)
        infer = EdgeInference(cfg, sch, torch.nn.Identity(), dec)

    if not stub:
        if args.pmc_child:
            native.set_substreams(1)  # counters per k_layer launch: launches must not share the device
        elif args.substreams is not None:
            native.set_substreams(args.substreams)
        substreams_max = native.set_substreams(0)  # (0 only queries)
        substreams = native.substreams_for(dec.dims(), B, T) if C["sampler"] != "none" else 1
    else:
        substreams_max = substreams = 1
    if C["sampler"] == "ddpm":
        # BASELINE config 5: the whole 1000-step ancestral sampler captured ONCE as a hipGraph; a step = one replay.  The start
        # noise is drawn into the graph's static input buffer inside the timed step.
        x_static = native.randn((B, T, cfg.n_mels), dev, seed=1)
        infer.sample_ddpm(sem, C["num_steps"], x_T=x_static, seed=3)  # eager warm-up: packs weights, sizes the workspace
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out_static = infer.sample_ddpm(sem, C["num_steps"], x_T=x_static, seed=3)

        def step():
            counter[0] += 1
            x_static.copy_(native.randn((B, T, cfg.n_mels), dev, seed=1000 + counter[0]))
            graph.replay()
            return out_static
    else:
        def local(sem_, n_, seed_, off_):
            if stub:  # stand-in sampler: rows that depend only on (seed, global utterance index), like the library's Philox noise
                rows = [torch.randn(T, cfg.n_mels, generator=torch.Generator().manual_seed(seed_ * 100003 + off_ + i)) for i in range(sem_.shape[0])]
                return torch.stack(rows).clamp_(-3, 3)
            return infer.generate_mel(sem_, n_, seed=seed_, batch_offset=off_)

        def step():
            counter[0] += 1
            seed = 1000 + counter[0]
            if world > 1 and micro > 1:
                Bm = B // micro
                slices = iter(range(micro))
                return generate_overlapped(lambda s_, n_, x_: local(s_, n_, seed, rank * B + next(slices) * Bm), sem, sem, C["num_steps"], world * B, micro)
            mel = local(sem, C["num_steps"], seed, rank * B)
            if world > 1:
                gather_out[0] = gather_batch(mel, world * B, out=gather_out[0])  # the one collective of the path (RCCL over xGMI)
                return gather_out[0]
            return mel

    for _ in range(warmup):
        step()
    class _HostEvent:  # stub mode: no device, so per-step times come from the host clock
        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    ev = [_HostEvent() if stub else torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        out = step()
        ev[i + 1].record()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cpu" if (rehearsal or stub) else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert out.shape == (world * B, T, cfg.n_mels) and bool(torch.isfinite(out[:2]).all())
    if args.pmc_child:
        return
    per_step = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    median_ms = per_step[len(per_step) // 2]
    if world > 1 and C["sampler"] == "ddim":
        # the all-gather alone, in its own loop (the headline loop above carries no instrumentation): device time of
        # gather_batch on an already computed shard, max over ranks
        mel = local(sem, C["num_steps"], 7, rank * B)
        for _ in range(5):
            sync()
            dist.barrier()
            g0, g1 = (_HostEvent(), _HostEvent()) if stub else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            g0.record()
            gather_out[0] = gather_batch(mel, world * B, out=gather_out[0])
            g1.record()
            sync()
            gather_ms.append(g0.elapsed_time(g1))
        gt = torch.tensor([sorted(gather_ms)[len(gather_ms) // 2]], device="cpu" if (rehearsal or stub) else dev, dtype=torch.float64)
        dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        gather_ms = float(gt.item())

    ms_per_step = dt / steps * 1e3
    frames_per_s = world * B * T / (dt / steps)
    H, M, L, W = cfg.hidden, cfg.n_mels, cfg.layers, cfg.attn_window_size
    what = (f"generate_mel {C['num_steps']}-step DDIM" if C["sampler"] == "ddim" else f"{C['num_steps']}-step DDPM sampler (one captured hipGraph)")
    metric = HEADLINE if args.config == 2 and (B, T) == (256, 512) else \
        f"mel-frames/sec (whole node), {what}, hidden={H} L={L}, B={B} T={T}"
    result = {
        "metric": metric, "value": frames_per_s, "unit": "mel-frames/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": ms_per_step, "ms_per_step_median": median_ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"BASELINE config {args.config}: {what}, decoder hidden={H} L={L} heads={cfg.heads} n_mels={M} window={W}, "
                               f"B={B}/GPU T={T} S={S}, synthetic weights + tokens, start noise drawn inside the timed call",
                   "batch_per_gpu": B, "frames": T, "sampler_steps": C["num_steps"],
                   "substreams": substreams,  # sub-batches (streams) this call shape is cut into (edtts.h: edtts_set_substreams; 1 = one piece)
                   "parallelism": (f"batch-sharded x{world}, all-gather of the final mel batch" + (f" overlapped with compute in {micro} slices" if micro > 1 else ""))
                   if world > 1 else "single GPU"},
        "mels_per_s": world * B / (dt / steps),
        "timed_region_s": dt,
    }
    if world > 1:
        result["rccl_ranks"] = rccl_ranks  # ranks that answered an all-reduce of ones on the collective backend
        result["collective_backend"] = dist.get_backend()
        if C["sampler"] == "ddim":
            result["allgather_ms"] = gather_ms  # median over 5, max over ranks, of the one all-gather per step (own loop)
            result["allgather_bytes"] = world * B * T * cfg.n_mels * 4
    if rehearsal:
        result["note"] = "REHEARSAL: ranks share one GPU, gloo collective -- not a scaling measurement"
    if stub:
        result["note"] = "STUB: stand-in CPU sampler over gloo (launcher / sharding / JSON contract test) -- not a measurement"
        result["data"] = "stub"
    total_flops = call_flops(H, L, M, S, T, W, B, C["num_steps"])
    peak = PEAK_TFLOPS[args.dtype]
    result["whole_call"] = {"algorithmic_tflop": total_flops / 1e12, "tflops": world * total_flops / (dt / steps) / 1e12,
                            "frac_of_mfma_peak": total_flops / (dt / steps) / 1e12 / peak, "peak_tflops": peak}

    if extras and not args.no_roofline and not stub:
        # ---- roofline leg: a SEPARATE loop with the per-launch HIP events on ----
        n_prof = 3 if C["sampler"] == "ddpm" else min(10, steps)
        per_call = C["num_steps"] * L
        if C["sampler"] == "ddpm":
            n_prof, per_call = 1, C["num_steps"] * L
        # the kernel is timed ALONE on the device (as rocprofv3's kernel trace of profiles/ sees it with EDTTS_SUBSTREAMS=1): with the
        # two-stream cut of the headline loop two launches share the SIMDs and a launch's own duration says nothing about its rate
        native.set_substreams(1)
        native.profile_enable(n_prof * per_call)
        for _ in range(n_prof):
            if C["sampler"] == "ddpm":
                infer.sample_ddpm(sem, C["num_steps"], x_T=x_static, seed=3)  # eager: events cannot be recorded inside a replay
            else:
                step()
        torch.cuda.synchronize(dev)
        (ms0, n0), (ms1, n1) = native.profile_collect()
        native.profile_enable(0)
        native.set_substreams(substreams_max)
        frames = B * T
        fl_layer = frames * sum(layer_flops_per_frame(H, M, S, T, W, l == L - 1) for l in range(L)) / L  # mean over the L launches
        avg = (ms0 + ms1) / max(n0, 1)
        ach = fl_layer / (avg * 1e-3) / 1e12
        result["roofline"] = {
            "bound": "mfma", "kernel": "k_layer (fused transformer layer: self-attn + cross-attn + FFN + QKV/DDIM tail)",
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "avg_launch_ms": avg, "launches_timed": n0,
            "algorithmic_gflop_per_launch": fl_layer / 1e9,
            "traffic": traffic["bytes"] if traffic else None, "traffic_read": traffic["read"] if traffic else None,
            "traffic_write": traffic["write"] if traffic else None, "traffic_source": traffic_src,
            "layer_kernels_share_of_step": avg * per_call / median_ms if C["sampler"] == "ddim" else None,
            "launch_mode": "one launch at a time (edtts_set_substreams(1)); the headline loop ran with substreams=%d" % substreams,
        }
        if traffic:
            gbs = traffic["bytes"] / (avg * 1e-3) / 1e9
            result["roofline"]["hbm"] = {"achieved_GBps": gbs, "peak_GBps": PEAK_HBM_GBS, "frac": gbs / PEAK_HBM_GBS}
    if extras and not stub and not args.no_cpu_baseline and C["sampler"] == "ddim" and args.config in (1, 2):
        result["cpu_baseline"] = cpu_baseline(cfg, sd, min(args.cpu_sample_batch, B), T, C["num_steps"])

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def cpu_baseline(cfg, sd, Bc, T, n_steps):
    """The CPU oracle (a PyTorch-CPU port of the reference path, validated against the reference's outputs in
    tests/test_oracle_vs_golden.py) timed on this host: one generate_mel on a bounded sample."""
    import torch
    from oracle import edtts_oracle as O
    # this process's CPU share on the GPU box is 16 cores (torch would otherwise start one thread per host core)
    threads = int(os.environ.get("EDTTS_CPU_THREADS", str(min(16, os.cpu_count() or 1))))
    torch.set_num_threads(threads)
    S = T // 2
    g = torch.Generator().manual_seed(2)
    sem = torch.randint(0, cfg.codebook_size, (Bc, S), generator=g)
    x_T = torch.randn(Bc, T, cfg.n_mels, generator=g)
    ab = O.schedule_tables(cfg.diff_steps)["alpha_bar"]
    reps = 1 if Bc > 8 else 20
    with torch.no_grad():
        O.generate_mel(sd, ab, sem[:2], x_T[:2], n_steps)  # warm-up
        t0 = time.perf_counter()
        for _ in range(reps):
            O.generate_mel(sd, ab, sem, x_T, n_steps)
        dt = (time.perf_counter() - t0) / reps
    return {"value": Bc * T / dt, "unit": "mel-frames/s", "cores": threads, "kind": "port", "cpu": cpu_model(),
            "sample": f"oracle/edtts_oracle.generate_mel, {n_steps}-step DDIM, B={Bc} T={T} fp32, torch {torch.__version__} CPU, "
                      f"{threads} threads, {reps} call(s) after warm-up, {dt:.2f} s per call", "seconds": dt}


if __name__ == "__main__":
    main()
