"""Data-parallel sharding of independent images over the GPUs of one node.

The reference is single-process / single-device; the hot path shards naturally
over images: one process per GPU, each rank runs its own step loop, and there are NO collectives
inside it.  The only exchanges are (1) one RCCL broadcast of the packed weight blob at load time
(xGMI is point-to-point, so one large contiguous transfer per peer instead of one per tensor) and
(2) an optional gather of the finished uint8 images.  ``torch.distributed`` backend "nccl" is
RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from .weights import blob_pack, blob_unpack

_CHUNK_ELEMS = 1 << 32  # bytes per broadcast call (4 GiB)


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, local_rank, world_size); a no-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_seeds(seeds: Sequence[int], rank: int, world: int) -> List[int]:
    """Contiguous partition: rank r gets seeds [r*n/world, (r+1)*n/world)."""
    n = len(seeds)
    lo, hi = (rank * n) // world, ((rank + 1) * n) // world
    return list(seeds[lo:hi])


def broadcast_weights(packed: Optional[Dict[str, torch.Tensor]], device, src: int = 0,
                      chunk_elems: int = _CHUNK_ELEMS, force: bool = False) -> Dict[str, torch.Tensor]:
    """Root packs its engine tensors into one byte blob; every rank receives blob + index and
    rebuilds zero-copy views.  One collective per ``chunk_elems`` bytes (4 GiB by default: the bf16 FLUX blob of
    23.8 GB goes out in 6 calls).  A single-rank run returns ``packed`` untouched unless ``force`` (an initialised process
    group of world size 1 then goes through the same pack / broadcast / unpack calls: how the single-GPU test box exercises
    the RCCL path, tests/test_gpu_dist.py)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        assert packed is not None
        return packed
    rank = dist.get_rank()
    if rank == src:
        blob, index = blob_pack(packed)
        meta = [index, blob.numel()]
    else:
        blob, meta = None, [None, None]
    dist.broadcast_object_list(meta, src=src)
    index, numel = meta
    if rank != src:
        blob = torch.empty(numel, dtype=torch.uint8, device=device)
    for off in range(0, numel, chunk_elems):
        dist.broadcast(blob[off:off + chunk_elems], src=src)
    return blob_unpack(blob, index)


def gather_images(u8: torch.Tensor, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Gather each rank's [n_img, H, W, 3] uint8 images on ``dst`` (None elsewhere).  ``shard_seeds`` hands ranks unequal
    image counts whenever the number of seeds is not a multiple of the world size (a rank may hold none), and a collective
    needs equal shapes: the per-rank counts are exchanged first, every rank pads its block to the largest count, and the
    padding is cut off on ``dst``."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [u8]
    world, rank = dist.get_world_size(), dist.get_rank()
    count = torch.tensor([u8.shape[0]], dtype=torch.int64, device=u8.device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    if n_max == 0:
        return [u8[:0] for _ in range(world)] if rank == dst else None
    block = u8
    if u8.shape[0] < n_max:
        block = torch.zeros(n_max, *u8.shape[1:], dtype=u8.dtype, device=u8.device)
        block[:u8.shape[0]] = u8
    block = block.contiguous()
    if dist.get_backend() == "nccl":
        out = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(out, block)  # RCCL has no gather-to-one primitive cheaper than this at 3 MiB/image
        return [o[:c] for o, c in zip(out, counts)] if rank == dst else None
    out = [torch.empty_like(block) for _ in range(world)] if rank == dst else None
    dist.gather(block, out, dst=dst)
    return [o[:c] for o, c in zip(out, counts)] if rank == dst else None
