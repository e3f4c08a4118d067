"""Multi-GPU layer: independent utterances are partitioned across ranks (one process per GPU); the only
collective on the data path is none at all -- weights are broadcast ONCE at start-up (RCCL over xGMI when
the backend is nccl) and generated ids are gathered at the end.  The reference is single-process /
single-device; this is new functionality with "run the reference B times" semantics.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def partition(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Static contiguous split [start, end) of n_items utterances."""
    base, rem = divmod(n_items, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_arena_host(model_dir: str) -> torch.Tensor:
    """Rank-local: read the safetensors checkpoint and build the arena bytes in host memory."""
    lib = _lib.load()
    n = C.c_uint64()
    if lib.q3a_arena_bytes(model_dir.encode(), C.byref(n)) != 0:
        raise RuntimeError((lib.q3a_last_error(None) or b"").decode())
    buf = torch.empty(n.value, dtype=torch.uint8)
    if lib.q3a_arena_pack(model_dir.encode(), C.c_void_p(buf.data_ptr()), n.value) != 0:
        raise RuntimeError((lib.q3a_last_error(None) or b"").decode())
    return buf


def arena_nbytes(model_dir: str) -> int:
    lib = _lib.load()
    n = C.c_uint64()
    if lib.q3a_arena_bytes(model_dir.encode(), C.byref(n)) != 0:
        raise RuntimeError((lib.q3a_last_error(None) or b"").decode())
    return int(n.value)


def broadcast_arena(model_dir: str, device: torch.device, src: int = 0) -> torch.Tensor:
    """Rank `src` packs the arena and every rank receives it with ONE broadcast of the whole byte range
    (a single large message: xGMI is point-to-point, a ring/tree of one big buffer is per-link bound).
    Works on CPU tensors with gloo (tests) and on GPU tensors with nccl == RCCL."""
    n = arena_nbytes(model_dir)
    if dist.get_rank() == src:
        t = pack_arena_host(model_dir).to(device)
    else:
        t = torch.empty(n, dtype=torch.uint8, device=device)
    dist.broadcast(t, src=src)
    return t


def gather_ids(local_ids: Sequence[Sequence[int]], n_total: int) -> List[List[int]]:
    """All ranks contribute their utterances' id lists; returns the global list in utterance order."""
    world = dist.get_world_size()
    objs: List[object] = [None] * world
    dist.all_gather_object(objs, [list(map(int, x)) for x in local_ids])
    out: List[List[int]] = []
    for r in range(world):
        out.extend(objs[r])  # partition() is contiguous and ordered by rank
    assert len(out) == n_total, (len(out), n_total)
    return out
