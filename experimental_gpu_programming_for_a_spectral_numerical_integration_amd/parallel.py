"""Multi-GPU sharding of the rod batch.

Rods (columns of the state panel) are independent and the operator is replicated, so the apply
needs NO collective: rank r owns one contiguous slab of columns.  The only exchange step is the
reassembly of results — an RCCL all-gather over xGMI (``torch.distributed`` backend "nccl" is RCCL
on ROCm) of either the whole output slab or just the integrated tip state of every column
(node 0 is the rod tip X = 1: chebyshev_differentiation.h:26 makes the nodes descend from L to 0).

One process per GPU.  Everything here is device-agnostic so that the same code is exercised on
CPU tensors with the gloo backend in tests.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_columns(total: int, world: int, rank: int, granule: int = 1) -> Tuple[int, int]:
    """Contiguous, balanced split of ``total`` columns: returns (first column, count) of ``rank``.

    ``granule`` keeps a rod's ``s`` state columns together (pass s = 4 for quaternion panels).
    The first ``(total/granule) % world`` ranks get one extra granule.
    """
    if world < 1 or not (0 <= rank < world) or granule < 1 or total % granule:
        raise ValueError("bad shard request")
    units = total // granule
    base, extra = divmod(units, world)
    mine = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first * granule, mine * granule


def gather_tip_state(Y_local: torch.Tensor, out: torch.Tensor = None, group=None, async_op: bool = False):
    """All-gather the tip-node value (node 0) of every local column.

    ``Y_local``: (M_local, rows) panel.  Returns ``(tips_all, work)`` where ``tips_all`` has
    M_local * world entries in rank order (equal shards assumed — weak scaling).
    """
    world = dist.get_world_size(group)
    tips = Y_local[..., 0].reshape(-1).contiguous()
    if out is None:
        out = torch.empty(tips.numel() * world, dtype=tips.dtype, device=tips.device)
    work = dist.all_gather_into_tensor(out, tips, group=group, async_op=async_op)
    return out, work


def gather_panel(Y_local: torch.Tensor, out: torch.Tensor = None, group=None, async_op: bool = False):
    """All-gather whole output slabs: with column-major panels every rank's slab is one contiguous
    block, so the gathered buffer IS the global (M_total, rows) panel (ncclAllGather in place
    layout)."""
    world = dist.get_world_size(group)
    flat = Y_local.reshape(-1, Y_local.shape[-1])
    if out is None:
        out = torch.empty((flat.shape[0] * world, flat.shape[1]), dtype=flat.dtype, device=flat.device)
    work = dist.all_gather_into_tensor(out, flat.contiguous(), group=group, async_op=async_op)
    return out, work
