"""Multi-process tests of the batch-sharding layer on CPU (gloo, world_size 2 and 3).  The local compute is a stand-in
(the product path has no CPU fallback): what is tested is the partition / gather index math and shard-count invariance."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from edge_diffusion_tts_amd.parallel import ShardedEdgeInference, gather_batch, shard_bounds, shard_sizes


def test_shard_bounds_partition():
    for total in (1, 7, 8, 256, 2048, 2051):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = shard_sizes(total, world)
            assert sum(sizes) == total and max(sizes) - min(sizes) <= 1
    assert shard_bounds(2048, 8, 3) == (768, 1024)  # BASELINE config 4: 256 utterances per GPU
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _fake_local_generate(sem, num_steps, x):
    # deterministic per-utterance function (depends only on that utterance's tokens and noise, like the real sampler);
    # like the real path (edtts_workspace_bytes rejects B < 1) it refuses an empty shard
    if sem.shape[0] == 0:
        raise RuntimeError("empty shard handed to the local sampler")
    return x * (1.0 + num_steps) + sem.float().mean(dim=1)[:, None, None]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(1)
        sem = torch.randint(0, 512, (total, 6), generator=g)
        x_T = torch.randn(total, 12, 5, generator=g)
        sh = ShardedEdgeInference(local_generate=_fake_local_generate)
        out = sh.generate_mel(sem, 4, x_T=x_T)
        ref = _fake_local_generate(sem, 4, x_T)
        ok = torch.equal(out, ref)
        # default noise path: drawn for the GLOBAL batch from the seed -> identical on every rank and for any world size
        out2 = sh.generate_mel(sem, 4, seed=11, n_mels=5)
        gg = torch.Generator().manual_seed(11)
        ref2 = _fake_local_generate(sem, 4, torch.randn(total, 12, 5, generator=gg))
        ok = ok and torch.equal(out2, ref2)
        # opt-in overlap of the all-gather with compute: identical result whenever the shards split evenly (else it falls back)
        for micro in (2, 4):
            sh_mb = ShardedEdgeInference(local_generate=_fake_local_generate, micro_batches=micro)
            ok = ok and torch.equal(sh_mb.generate_mel(sem, 4, x_T=x_T), ref)
        # raw gather with ragged shards
        lo, hi = shard_bounds(total, world, rank)
        ok = ok and torch.equal(gather_batch(ref[lo:hi].clone(), total), ref)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,total", [(2, 8), (2, 7), (3, 10), (2, 16), (3, 2), (2, 1)])
def test_sharded_generate_gloo(world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    results = dict(q.get(timeout=5) for _ in range(world))
    assert results == {r: True for r in range(world)}


def _run_bench(extra_env, *argv):
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EDTTS_BENCH_STUB="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus2_starts_itself():
    """`python bench.py --gpus 2` with NO outer torchrun (the driver's command form): the parent spawns one rank per GPU before
    touching any device and relays rank 0's line.  Stub sampler + gloo here (no GPU in this container): the launcher, the
    sharding, the all-gather, the max-over-ranks timing and the JSON contract are what is exercised."""
    d = _run_bench({}, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--frames", "32")
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["unit"] == "mel-frames/s"
    assert d["config"]["batch_per_gpu"] == 4 and d["allgather_bytes"] == 2 * 4 * 32 * 80 * 4 and d["allgather_ms"] > 0
    assert abs(d["value"] - 2 * 4 * 32 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]  # whole-job aggregate


def test_bench_spawned_rank_failure_ends_the_job():
    """A rank that dies must take the job down with a non-zero status, not leave the others waiting in a collective."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EDTTS_BENCH_STUB="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    # config 5 is refused by the stub on every rank -> all children exit non-zero
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--config", "5", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


def test_bench_parent_interrupted_takes_its_ranks_down():
    """The launcher parent killed (driver timeout, Ctrl-C) must not leave N ranks running on the GPUs: SIGTERM to the parent ends
    the exact children it started.  Stub ranks that would run for a long time; the parent is terminated once they are up."""
    import signal
    import subprocess
    import sys
    import time
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EDTTS_BENCH_STUB="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "200000", "--warmup", "1", "--batch", "4",
                          "--frames", "32"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    def children():
        out = subprocess.run(["ps", "-o", "pid=", "--ppid", str(p.pid)], capture_output=True, text=True).stdout.split()
        return [int(x) for x in out]

    deadline = time.time() + 120
    kids = []
    while time.time() < deadline and len(kids) < 2:
        time.sleep(0.5)
        kids = children()
    assert len(kids) == 2, "the two ranks did not start"
    time.sleep(2.0)
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=60) != 0
    time.sleep(0.5)
    for k in kids:
        alive = subprocess.run(["ps", "-p", str(k), "-o", "stat="], capture_output=True, text=True).stdout.strip()
        assert alive == "" or alive.startswith("Z"), f"rank process {k} survived its parent ({alive})"
