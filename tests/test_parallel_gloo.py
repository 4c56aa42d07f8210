"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: column sharding + all-gather reassembly.
The apply itself has no CPU path, so the per-rank compute is stood in by the oracle (the checker)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from experimental_gpu_programming_for_a_spectral_numerical_integration_amd import parallel


def test_shard_columns_partitions_exactly():
    for total, world, gran in [(16, 2, 4), (4 * 65536 * 8, 8, 4), (28, 3, 4), (7, 4, 1), (0, 2, 1), (12, 5, 4)]:
        spans = [parallel.shard_columns(total, world, r, gran) for r in range(world)]
        assert spans[0][0] == 0
        for (a, n), (b, _) in zip(spans, spans[1:]):
            assert a + n == b
        assert spans[-1][0] + spans[-1][1] == total
        counts = [n for _, n in spans]
        assert all(c % gran == 0 for c in counts) and max(counts) - min(counts) <= gran
    with pytest.raises(ValueError):
        parallel.shard_columns(10, 2, 0, 4)
    with pytest.raises(ValueError):
        parallel.shard_columns(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, rods, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle

        A = oracle.dn(n)
        M = 4 * rods
        rng = np.random.default_rng(0x5EED)
        Xg = rng.uniform(-1, 1, size=(M, n))                       # the global panel, identical on every rank
        first, count = parallel.shard_columns(M, world, rank, granule=4)
        Xl = Xg[first:first + count]
        Yl = torch.from_numpy(np.ascontiguousarray(oracle.apply(A, np.asfortranarray(Xl.T)).T))
        counts = parallel.shard_counts(M, world, granule=4)
        ragged = len(set(counts)) > 1
        tips, _ = parallel.gather_tip_state(Yl, columns_per_rank=counts)
        full, _ = parallel.gather_panel(Yl, columns_per_rank=counts)
        if ragged:
            # ragged shards without the per-rank counts must be refused up front, not hang in the collective
            tips_async = tips
            out = torch.empty(M, dtype=torch.float64)
            t2, _ = parallel.gather_tip_state(Yl, out=out, columns_per_rank=counts)
            assert t2 is out and torch.equal(out, tips)
        else:
            tips_async, work = parallel.gather_tip_state(Yl, async_op=True)
            work.wait()
        if rank == 0:
            Yg = oracle.apply(A, np.asfortranarray(Xg.T)).T
            ok = (np.array_equal(full.numpy(), Yg) and np.array_equal(tips.numpy(), Yg[:, 0])
                  and np.array_equal(tips_async.numpy(), Yg[:, 0]))
            q.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rods", [6, 7])          # 7 rods over 2 ranks: ragged shards (4 + 3 rods)
def test_sharded_apply_reassembles_over_gloo(rods):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 16, rods, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_ragged_gather_needs_counts():
    assert parallel.shard_counts(28, 2, granule=4) == [16, 12]
    assert parallel.shard_counts(24, 2, granule=4) == [12, 12]
