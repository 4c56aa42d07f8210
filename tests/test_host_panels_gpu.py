"""Host panels through the C ABI (sri_apply_host / sri_integrate_host — what the drop-in header calls).

The panel crosses PCIe in blocks: CPU threads copy it into and out of pinned staging slots owned by the handle, three streams
carry the blocks in / apply them / bring them back (csrc/sri_host_panels.hip; the stage order itself is checked on the CPU
against a mock card, tests/test_host_pipeline.py).  The result must be the device path's, bit for bit, whatever the blocking:
several blocks, ragged last block, strided panels, odd row counts, in place, growing sizes — with ordinary pageable numpy
memory on the caller's side and ordinary pageable torch transfers next to it in the same process, under the process's
default allocator settings (round 3 ran this file with malloc trimming disabled; nothing here depends on that any more:
the library no longer page-locks, registers or hands to HIP any caller memory).
"""
import numpy as np
import pytest

from conftest import uniform_panel

pytestmark = pytest.mark.gpu


def _device_apply(op, X):
    import torch

    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()          # (M, rows) contiguous = column-major rows x M
    return np.asfortranarray(op.apply(Xd).cpu().numpy().T)


def _device_integrate(op, B, xb):
    import torch

    Bd = torch.from_numpy(np.ascontiguousarray(B.T)).cuda()
    return np.asfortranarray(op.integrate(Bd, torch.from_numpy(xb).cuda()).cpu().numpy().T)


@pytest.mark.parametrize("n,which,M", [
    (64, 0, 50_000),        # 512-byte columns: 4 MiB blocks of 8 192 columns -> 7 blocks, ragged last one
    (256, 0, 20_001),       # 5 blocks, odd column count
    (16, 0, 777),           # one block
    (256, 1, 9_000),        # Dn_NN^-1, 255 rows: odd leading dimension on the device
    (64, 2, 10_000),        # Dn_NN, 63 rows
    (1024, 0, 2_500),       # ascending-k handle
])
def test_host_panel_equals_device_panel(sri, n, which, M):
    op = sri.SpectralApply(n, which)
    X = uniform_panel(op.rows, M)
    Y = op.apply_host(X)
    assert Y.flags.f_contiguous and Y.shape == (op.rows, M)
    assert np.array_equal(Y, _device_apply(op, X))


def test_host_panel_blocks_are_reused_and_grow(sri):
    """One handle, three calls of different size: the slot ring is reused, regrown, and never leaks a previous result."""
    op = sri.SpectralApply(64, 0)
    for M, seed in ((3_000, 1), (70_000, 2), (129, 3)):
        X = uniform_panel(64, M, seed)
        assert np.array_equal(op.apply_host(X), _device_apply(op, X)), M


def test_strided_host_panels_keep_their_padding(sri):
    """X and Y are the leading rows of taller matrices: the rows beyond are not read into the result and not written."""
    n, M, ldx, ldy = 64, 40_000, 70, 67
    op = sri.SpectralApply(n, 0)
    big_x = np.asfortranarray(np.full((ldx, M), np.nan))
    big_x[:n] = uniform_panel(n, M)
    big_y = np.asfortranarray(np.full((ldy, M), -7.25))
    out = op.apply_host(big_x[:n], out=big_y[:n])
    assert out.base is big_y or out is big_y[:n] or np.shares_memory(out, big_y)
    assert np.array_equal(big_y[:n], _device_apply(op, np.asfortranarray(big_x[:n])))
    assert np.all(big_y[n:] == -7.25)


def test_host_panel_in_place(sri):
    n, M = 128, 30_000
    op = sri.SpectralApply(n, 0)
    X = uniform_panel(n, M)
    want = _device_apply(op, X)
    buf = X.copy(order="F")
    op.apply_host(buf, out=buf)
    assert np.array_equal(buf, want)


@pytest.mark.parametrize("n,M", [(256, 12_000), (64, 45_000), (17, 300), (257, 40_000), (129, 70_001)])   # (257, 129: the term as the store stage's fused multiply-add, operands staged from the device slot)
def test_integrate_host_equals_device(sri, n, M):
    op = sri.SpectralApply(n, sri.OP_DNN_INV)
    B = uniform_panel(n - 1, M, 11)
    xb = np.random.default_rng(12).uniform(-1, 1, M)
    assert np.array_equal(op.integrate_host(B, xb), _device_integrate(op, B, xb))


def test_host_panel_argument_errors(sri):
    op = sri.SpectralApply(64, 0)
    with pytest.raises(ValueError):
        op.apply_host(np.zeros((63, 4), order="F"))
    with pytest.raises(ValueError):
        op.apply_host(np.zeros((64, 4), order="F"), out=np.zeros((64, 5), order="F"))
    with pytest.raises(ValueError):
        op.apply_host(np.zeros((64, 4), order="F"), out=np.zeros((64, 4), order="C"))
    with pytest.raises(ValueError):
        sri.SpectralApply(64, sri.OP_DNN_INV).integrate_host(np.zeros((63, 4), order="F"), np.zeros(3))
    assert op.apply_host(np.zeros((64, 0), order="F")).shape == (64, 0)


def test_read_only_host_panel(sri):
    """A read-only input (numpy write flag off) is only ever read — by the CPU — on its way into the staging slots."""
    n, M = 64, 20_000
    op = sri.SpectralApply(n, 0)
    X = uniform_panel(n, M)
    Xro = X.copy(order="F")
    Xro.setflags(write=False)
    assert np.array_equal(op.apply_host(Xro), _device_apply(op, X))


def test_a_failed_hip_call_of_the_caller_is_not_reported_as_a_failed_launch(sri):
    """HIP keeps the last error per thread: a launch must report its own status, not the application's earlier failure
    (torch itself raises on such a leftover, so everything torch does here happens before it, or after it is cleared)."""
    import ctypes

    import torch

    path = next(line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line)
    hip = ctypes.CDLL(path)                                           # the runtime this process already uses
    op = sri.SpectralApply(64, 0)
    integ = sri.SpectralApply(64, sri.OP_DNN_INV)
    Xs = torch.rand((512, 64), dtype=torch.float64, device="cuda")    # one-tile kernel
    Xl = torch.rand((40_000, 64), dtype=torch.float64, device="cuda")  # general kernel
    B = torch.rand((1_000, 63), dtype=torch.float64, device="cuda")
    xb = torch.rand((1_000,), dtype=torch.float64, device="cuda")
    Ys, Yl, Yi = torch.empty_like(Xs), torch.empty_like(Xl), torch.empty_like(B)
    want = (op.apply(Xs), op.apply(Xl), integ.integrate(B, xb))
    torch.cuda.synchronize()
    junk = (ctypes.c_char * 4096)()
    try:
        assert hip.hipHostUnregister(junk) != 0                       # the caller's own failed call
        op.apply(Xs, out=Ys)
        op.apply(Xl, out=Yl)
        integ.integrate(B, xb, out=Yi)
        host = op.apply_host(np.asfortranarray(np.ones((64, 3000))))
    finally:
        hip.hipGetLastError()                                         # what an application would do; torch needs it
    torch.cuda.synchronize()
    assert torch.equal(Ys, want[0]) and torch.equal(Yl, want[1]) and torch.equal(Yi, want[2])
    assert np.isfinite(host).all()


def test_two_threads_on_adjacent_slabs_and_a_shared_input(sri):
    """The documented sharding — one host thread and one handle per slab — on ONE host allocation: the slabs are adjacent
    (they share a boundary page: 255-row columns are 2 040 bytes), and a second pair of threads reads the SAME input.
    Each handle owns its staging ring and copy threads; the caller's memory is shared between the calls but only touched by
    CPU copies, so a call that finishes early has nothing to release under another one.  Repeated, against the device path."""
    import threading

    n, which, M = 256, 1, 12_001                                   # 255 rows: nothing is page aligned
    ops = [sri.SpectralApply(n, which) for _ in range(4)]
    rows = ops[0].rows
    X = uniform_panel(rows, 2 * M, seed=11)                         # one allocation, two slabs of M columns
    Y = np.asfortranarray(np.full((rows, 2 * M), np.nan))
    Z = [np.asfortranarray(np.full((rows, M), np.nan)) for _ in range(2)]
    ref = _device_apply(ops[0], X)
    errors = []

    def work(op, xin, yout):
        try:
            for _ in range(3):
                yout[...] = np.nan
                op.apply_host(xin, out=yout)
                if np.isnan(yout).any():
                    errors.append("NaN left in the output")
        except Exception as e:                                      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(ops[0], X[:, :M], Y[:, :M])),
               threading.Thread(target=work, args=(ops[1], X[:, M:], Y[:, M:])),
               threading.Thread(target=work, args=(ops[2], X[:, :M], Z[0])),      # the same input slab as thread 0, read-only
               threading.Thread(target=work, args=(ops[3], X[:, :M], Z[1]))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    assert np.array_equal(Y, ref)
    assert np.array_equal(Z[0], ref[:, :M]) and np.array_equal(Z[1], ref[:, :M])


def test_pageable_transfers_of_the_caller_next_to_host_panel_calls(sri):
    """What a C++ host does next to sri_apply_host (and what round 3's page-locking could not live with): its own large
    pageable copies from and to freshly allocated, freed and recycled heap buffers — the SAME buffers the library is
    handed, and neighbours that share their first and last pages — interleaved with host-panel calls.  Default allocator."""
    import torch

    n, which, M = 256, 1, 16_385                                    # 255 rows: 2 040-byte columns, nothing page aligned
    op = sri.SpectralApply(n, which)
    rows = op.rows
    ref_op = sri.SpectralApply(n, which)
    for rep in range(6):
        big = uniform_panel(rows, 3 * M, seed=100 + rep)           # fresh heap memory every round (freed at the end of it)
        X = big[:, M:2 * M]                                         # the middle slab: both neighbours share its boundary pages
        left = torch.from_numpy(np.ascontiguousarray(big[:, :M].T)).cuda()          # pageable copy of the left neighbour
        Y = np.asfortranarray(np.full((rows, M), np.nan))
        op.apply_host(X, out=Y)
        same = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()                   # pageable copy of the very buffer just used
        right = torch.from_numpy(np.ascontiguousarray(big[:, 2 * M:].T)).cuda()
        back = ref_op.apply(same).cpu().numpy()                                     # pageable copy back into fresh memory
        assert np.array_equal(Y, np.asfortranarray(back.T)), rep
        assert np.array_equal(left.cpu().numpy().T, big[:, :M]) and np.array_equal(right.cpu().numpy().T, big[:, 2 * M:])
        del big, X, Y, left, same, right, back


def test_copy_thread_settings_give_the_same_bits(sri):
    n, M = 256, 30_001
    op = sri.SpectralApply(n, 0)
    X = uniform_panel(n, M, seed=5)
    want = _device_apply(op, X)
    for threads in (0, 1, 3, -1):
        op.set_host_copy_threads(threads)
        assert np.array_equal(op.apply_host(X), want), threads
    with pytest.raises(RuntimeError):
        op.set_host_copy_threads(1000)


def test_caller_pinned_memory_is_used_in_place_and_the_claim_is_checked(sri):
    """sri_apply_set_host_memory_pinned: a caller whose panels are page-locked (torch's pinned allocator here, hipHostMalloc
    in C++) skips the staging copies; a panel that is NOT page-locked is refused instead of being handed to the copy engines."""
    import torch

    n, M = 256, 40_000
    op = sri.SpectralApply(n, sri.OP_DNN_INV)
    rows = op.rows
    X = uniform_panel(rows, M, seed=9)
    xb = np.random.default_rng(10).uniform(-1, 1, M)
    want = _device_integrate(op, X, xb)
    Xp = torch.from_numpy(np.ascontiguousarray(X.T)).pin_memory()
    Yp = torch.empty_like(Xp).pin_memory()
    bp = torch.from_numpy(xb).pin_memory()
    op.set_host_memory_pinned(True)
    out = op.integrate_host(Xp.numpy().T, bp.numpy(), out=Yp.numpy().T)
    assert np.array_equal(out, want)
    with pytest.raises(RuntimeError):
        op.integrate_host(X, xb)                                    # plain numpy memory: the promise does not hold
    op.set_host_memory_pinned(False)
    assert np.array_equal(op.integrate_host(X, xb), want)           # back on the staged path


def test_a_pinned_claim_is_checked_over_the_whole_range(sri):
    """Round 5 (advice): the caller's promise of page-locked panels is verified for EVERY byte, not for the two ends — a panel
    that spans two registered buffers with a pageable gap between them, or a registration shorter than the panel, is refused
    (SRI_EINVAL) instead of being handed to a copy engine; the same memory registered as a whole is accepted."""
    import ctypes
    import torch

    n, M = 64, 6_000
    op = sri.SpectralApply(n, sri.OP_DN)
    rows = op.rows
    X = uniform_panel(rows, M, seed=21)
    want = op.apply_host(X)
    Y = np.empty_like(X)
    rt = torch.cuda.cudart()
    page = 4096
    raw = np.empty(X.nbytes + 2 * page, dtype=np.uint8)
    off = (-raw.ctypes.data) % page
    buf = raw[off:off + X.nbytes].view(np.float64).reshape(X.shape[::-1]).T        # page-aligned column-major rows x M
    assert buf.ctypes.data % page == 0 and buf.flags.f_contiguous
    buf[...] = X
    Yraw = np.empty(X.nbytes + 2 * page, dtype=np.uint8)
    yoff = (-Yraw.ctypes.data) % page
    Yb = Yraw[yoff:yoff + X.nbytes].view(np.float64).reshape(X.shape[::-1]).T
    ybytes = (X.nbytes + page - 1) // page * page
    assert int(rt.cudaHostRegister(Yb.ctypes.data, ybytes, 0)) == 0
    third = (X.nbytes // 3) // page * page
    op.set_host_memory_pinned(True)
    try:
        # two registered thirds with a pageable one between them
        assert int(rt.cudaHostRegister(buf.ctypes.data, third, 0)) == 0
        assert int(rt.cudaHostRegister(buf.ctypes.data + 2 * third, ybytes - 2 * third, 0)) == 0
        with pytest.raises(RuntimeError):
            op.apply_host(buf, out=Yb)
        assert int(rt.cudaHostUnregister(buf.ctypes.data + 2 * third)) == 0
        # a registration shorter than the panel
        with pytest.raises(RuntimeError):
            op.apply_host(buf, out=Yb)
        assert int(rt.cudaHostUnregister(buf.ctypes.data)) == 0
        # the whole panel registered: accepted, same bits
        assert int(rt.cudaHostRegister(buf.ctypes.data, ybytes, 0)) == 0
        out = op.apply_host(buf, out=Yb)
        assert np.array_equal(out, want)
        assert int(rt.cudaHostUnregister(buf.ctypes.data)) == 0
    finally:
        op.set_host_memory_pinned(False)
        rt.cudaHostUnregister(Yb.ctypes.data)
    del ctypes


def test_host_staging_can_be_given_back(sri):
    """sri_apply_host_trim releases the pinned staging ring and the device slots a handle grew for a large panel (round 5
    advice: the pipe never shrinks by itself); the next host call allocates what it needs and gives the same bits."""
    op = sri.SpectralApply(128, sri.OP_DN)
    X = uniform_panel(op.rows, 30_000, seed=4)
    a = op.apply_host(X)
    op.host_trim()
    op.host_trim()                                                   # idempotent
    small = np.asfortranarray(X[:, :500])
    assert np.array_equal(op.apply_host(small), a[:, :500])
    assert np.array_equal(op.apply_host(X), a)

