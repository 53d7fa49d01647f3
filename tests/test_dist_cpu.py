"""world_size-2 gloo tests of the data-parallel plumbing (no GPU): seed sharding, the single
weight-blob broadcast and the image gather."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffusionkit_amd.dist import shard_seeds


def test_shard_seeds_partitions():
    seeds = list(range(64))
    parts = [shard_seeds(seeds, r, 8) for r in range(8)]
    assert parts[3] == list(range(24, 32))
    assert sum(parts, []) == seeds
    parts = [shard_seeds(list(range(10)), r, 4) for r in range(4)]
    assert sum(parts, []) == list(range(10)) and max(map(len, parts)) - min(map(len, parts)) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from diffusionkit_amd import dist as dk
    from diffusionkit_amd.config import tiny_flux
    from diffusionkit_amd.weights import pack_mmdit, synth_mmdit_weights
    r, lr, w = dk.init_distributed("gloo")
    cfg = tiny_flux(1, 1)
    packed = pack_mmdit(cfg, synth_mmdit_weights(cfg, seed=99), "cpu") if r == 0 else None
    got = dk.broadcast_weights(packed, "cpu", src=0)
    ref = pack_mmdit(cfg, synth_mmdit_weights(cfg, seed=99), "cpu")
    ok = set(got) == set(ref) and all(torch.equal(got[k], ref[k]) for k in ref)
    # the chunked form the full-size blob takes (several collectives, a ragged last chunk)
    got = dk.broadcast_weights(packed, "cpu", src=0, chunk_elems=100_003)
    ok = ok and set(got) == set(ref) and all(torch.equal(got[k], ref[k]) for k in ref)
    imgs = torch.full((2, 4, 4, 3), r, dtype=torch.uint8)
    g = dk.gather_images(imgs, dst=0)
    if r == 0:
        ok = ok and len(g) == w and all(int(g[i].max()) == i for i in range(w))
    else:
        ok = ok and g is None
    # uneven shards (3 seeds over 2 ranks -> 1 + 2 images; 1 seed -> rank 0 holds none): counts exchanged, blocks padded
    for n_seeds in (3, 1):
        mine = dk.shard_seeds(list(range(n_seeds)), r, w)
        imgs = torch.stack([torch.full((4, 4, 3), 10 + s, dtype=torch.uint8) for s in mine]) if mine else torch.zeros(0, 4, 4, 3, dtype=torch.uint8)
        g = dk.gather_images(imgs, dst=0)
        if r == 0:
            flat = torch.cat(g, 0)
            ok = ok and [int(t.shape[0]) for t in g] == [len(dk.shard_seeds(list(range(n_seeds)), i, w)) for i in range(w)]
            ok = ok and flat.shape[0] == n_seeds and [int(flat[i].max()) for i in range(n_seeds)] == [10 + s for s in range(n_seeds)]
        else:
            ok = ok and g is None
    seeds = dk.shard_seeds(list(range(8)), r, w)
    ok = ok and seeds == list(range(4 * r, 4 * r + 4))
    dist.barrier()
    q.put((r, ok))
    dist.destroy_process_group()


def test_broadcast_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}


def test_bench_multi_rank_control_path_dry_run():
    """`bench.py --gpus 2 --batch auto --dry-run`: the script's own rank spawn, rendezvous, weight broadcast, barriers, MAX over ranks
    and per-rank gather through gloo on the host (the first 8-GPU run must not be the first run of this code)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "auto", "--steps", "3", "--warmup", "1", "--dry-run"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["world"] == 2 and d["collective_backend"] == "gloo" and d["scaling"] == "weak"
    assert "batch 8 images per rank" in d["config"]["workload"] and d["config"]["parallelism"].startswith("dp2")
    assert len(d["per_rank_images_per_s"]) == 2 and d["per_rank_images_per_s"][0] > d["per_rank_images_per_s"][1]  # rank 1 sleeps longer
    # value = all ranks' images / the slowest rank's time (MAX over ranks), not the sum of the per-rank rates
    assert abs(d["value"] - 2 * 3 * 8 / (d["ms_per_step"] * 3e-3)) <= 1e-2 * d["value"]
    assert d["value"] <= 2 * d["per_rank_images_per_s"][1] * 1.02
    assert d["weight_blob_gb"] > 0
