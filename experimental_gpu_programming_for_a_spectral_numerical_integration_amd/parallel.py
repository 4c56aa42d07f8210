"""Multi-GPU sharding of the rod batch — Python mirror of the C ABI's communicator.

Rods (columns of the state panel) are independent and the operator is replicated, so the apply
needs NO collective: rank r owns one contiguous slab of columns (``shard_columns``).  The only
exchange step is the reassembly of results — an RCCL all-gather over xGMI of either the
integrated tip state of every column (node 0 is the rod tip X = 1: chebyshev_differentiation.h:26
makes the nodes descend from L to 0) or the whole output slab.

The exchange lives in ``libsri_hip.so`` (``csrc/sri_comm.hip``: ``sri_comm_create``,
``sri_allgather_tips``, ``sri_allgather_panel`` — RCCL called from C++ on the caller's HIP stream,
so a C++ host program needs neither Python nor torch); ``Communicator`` binds it.  One process
per GPU.  ``gather_tip_state`` / ``gather_panel`` route CUDA tensors through a ``Communicator``
and CPU tensors through ``torch.distributed`` (gloo), which is how the sharding logic is tested
without GPUs.  Ragged shards (rods % world != 0) are supported on both routes.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import SriError, _chk, _vp, load_library

COMM_ID_BYTES = 128


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


def shard_counts(total: int, world: int, granule: int = 1) -> List[int]:
    """Column count of every rank under ``shard_columns`` (the ``columns_per_rank`` of the C ABI)."""
    return [shard_columns(total, world, r, granule)[1] for r in range(world)]


class Communicator:
    """``sri_comm_t``: an RCCL communicator owned by libsri_hip.so, one rank per GPU.

    ``Communicator(id, rank, world, device)`` is collective over the ``world`` ranks; ``id`` is the
    128-byte token of ``Communicator.unique_id()`` made on ONE rank and shipped to the others by
    any side channel.  ``Communicator.from_torch_distributed()`` uses an initialised
    ``torch.distributed`` group as that side channel (the token is broadcast as an object, so it
    works over gloo and RCCL alike).
    """

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {COMM_ID_BYTES} bytes")
        self._h = _vp()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        _chk(load_library().sri_comm_create(C.byref(self._h), buf, self.rank, self.world, self.device), "sri_comm_create")

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _chk(load_library().sri_comm_unique_id(buf), "sri_comm_unique_id")
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, device: int, group=None) -> "Communicator":
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], rank, world, device)

    def _counts(self, counts: Optional[Sequence[int]], local: int):
        if counts is None:
            return None, local * self.world
        if len(counts) != self.world or int(counts[self.rank]) != local:
            raise ValueError("columns_per_rank must have one entry per rank and match the local shard")
        return (C.c_size_t * self.world)(*[int(c) for c in counts]), int(sum(counts))

    def allgather_tips(self, tips: torch.Tensor, out: Optional[torch.Tensor] = None,
                       columns_per_rank: Optional[Sequence[int]] = None, stream: Optional[int] = None) -> torch.Tensor:
        """``sri_allgather_tips`` on torch's current stream (or ``stream``, a raw hipStream_t)."""
        if not (tips.is_cuda and tips.dtype == torch.float64 and tips.is_contiguous() and tips.device.index == self.device):
            raise TypeError("tips must be a contiguous float64 CUDA tensor on the communicator's device")
        carr, total = self._counts(columns_per_rank, tips.numel())
        if out is None:
            out = torch.empty(total, dtype=torch.float64, device=tips.device)
        elif not (out.is_cuda and out.dtype == torch.float64 and out.is_contiguous() and out.numel() == total):
            raise ValueError(f"out must be a contiguous float64 CUDA tensor of {total} entries")
        s = torch.cuda.current_stream(tips.device).cuda_stream if stream is None else stream
        _chk(load_library().sri_allgather_tips(self._h, tips.data_ptr(), tips.numel(), out.data_ptr(), carr, s),
             "sri_allgather_tips")
        return out

    def allgather_panel(self, Y: torch.Tensor, out: Optional[torch.Tensor] = None,
                        columns_per_rank: Optional[Sequence[int]] = None, stream: Optional[int] = None) -> torch.Tensor:
        """``sri_allgather_panel``: ``Y`` is a (columns_local, rows) panel; returns the global (columns, rows) panel."""
        if not (Y.is_cuda and Y.dtype == torch.float64 and Y.is_contiguous() and Y.device.index == self.device):
            raise TypeError("Y must be a contiguous float64 CUDA tensor on the communicator's device")
        rows = Y.shape[-1]
        cols = Y.numel() // rows
        carr, total = self._counts(columns_per_rank, cols)
        if out is None:
            out = torch.empty((total, rows), dtype=torch.float64, device=Y.device)
        elif not (out.is_cuda and out.dtype == torch.float64 and out.is_contiguous() and out.numel() == total * rows):
            raise ValueError(f"out must be a contiguous float64 CUDA tensor of {total} x {rows}")
        s = torch.cuda.current_stream(Y.device).cuda_stream if stream is None else stream
        _chk(load_library().sri_allgather_panel(self._h, Y.data_ptr(), rows, cols, out.data_ptr(), carr, s),
             "sri_allgather_panel")
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load_library().sri_comm_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _gather_flat(flat: torch.Tensor, counts: Optional[Sequence[int]], group, async_op: bool, out: torch.Tensor = None):
    """torch.distributed route (CPU tensors / gloo in tests): ragged shards are padded to the largest
    one for the collective and trimmed afterwards, so every rank contributes equally."""
    world = dist.get_world_size(group)
    if counts is None or len(set(int(c) for c in counts)) == 1:
        if out is None:
            out = torch.empty(flat.numel() * world, dtype=flat.dtype, device=flat.device)
        work = dist.all_gather_into_tensor(out.view(-1), flat, group=group, async_op=async_op)
        return out, work
    if len(counts) != world or int(counts[dist.get_rank(group)]) != flat.numel():
        raise ValueError("counts must have one entry per rank and match the local shard")
    if async_op:
        raise ValueError("async_op is not available for ragged shards (the result is trimmed after the wait)")
    cap = int(max(counts))
    padded = torch.zeros(cap, dtype=flat.dtype, device=flat.device)
    padded[:flat.numel()] = flat
    buf = torch.empty(cap * world, dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    res = torch.cat([buf[r * cap:r * cap + int(counts[r])] for r in range(world)])
    if out is not None:
        out.view(-1).copy_(res)
        res = out
    return res, None


def gather_tip_state(Y_local: torch.Tensor, out: torch.Tensor = None, group=None, async_op: bool = False,
                     columns_per_rank: Optional[Sequence[int]] = None, comm: Optional[Communicator] = None):
    """All-gather the tip-node value (node 0) of every local column.

    ``Y_local``: (M_local, rows) panel.  Returns ``(tips_all, work)``; ``tips_all`` holds every
    rank's columns in rank order.  ``columns_per_rank`` (``shard_counts``) is required when the
    shards are ragged.  CUDA panels go through ``comm`` (the C ABI, RCCL); CPU panels through
    ``torch.distributed``.
    """
    tips = Y_local[..., 0].reshape(-1).contiguous()
    if comm is not None:
        return comm.allgather_tips(tips, out=out, columns_per_rank=columns_per_rank), None
    return _gather_flat(tips, columns_per_rank, group, async_op, out)


def gather_panel(Y_local: torch.Tensor, out: torch.Tensor = None, group=None, async_op: bool = False,
                 columns_per_rank: Optional[Sequence[int]] = None, comm: Optional[Communicator] = None):
    """All-gather whole output slabs: with column-major panels every rank's slab is one contiguous
    block, so the gathered buffer IS the global (M_total, rows) panel (ncclAllGather in place
    layout)."""
    flat = Y_local.reshape(-1, Y_local.shape[-1]).contiguous()
    rows = flat.shape[1]
    if comm is not None:
        return comm.allgather_panel(flat, out=out, columns_per_rank=columns_per_rank), None
    counts = None if columns_per_rank is None else [int(c) * rows for c in columns_per_rank]
    res, work = _gather_flat(flat.reshape(-1), counts, group, async_op, out)
    return res.reshape(-1, rows), work


__all__ = ["shard_columns", "shard_counts", "Communicator", "gather_tip_state", "gather_panel", "SriError"]
