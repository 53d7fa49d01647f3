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

_CHUNK_ELEMS = 1 << 31  # 4 GiB of bf16 per broadcast call


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
                      chunk_elems: int = _CHUNK_ELEMS) -> Dict[str, torch.Tensor]:
    """Root packs its engine tensors into one bf16 blob; every rank receives blob + index and
    rebuilds zero-copy views.  One collective per ``chunk_elems`` elements (4 GiB by default: the FLUX blob of
    11.9 G elements goes out in 6 calls)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
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
        blob = torch.empty(numel, dtype=torch.bfloat16, device=device)
    for off in range(0, numel, chunk_elems):
        dist.broadcast(blob[off:off + chunk_elems], src=src)
    return blob_unpack(blob, index)


def gather_images(u8: torch.Tensor, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Gather each rank's [n_img, H, W, 3] uint8 images on ``dst`` (None elsewhere)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [u8]
    world = dist.get_world_size()
    if dist.get_backend() == "nccl":
        out = [torch.empty_like(u8) for _ in range(world)]
        dist.all_gather(out, u8)  # RCCL has no gather-to-one primitive cheaper than this at 3 MiB/image
        return out if dist.get_rank() == dst else None
    out = [torch.empty_like(u8) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(u8, out, dst=dst)
    return out
