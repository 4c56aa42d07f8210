"""Host panels through the C ABI (sri_apply_host / sri_integrate_host — what the drop-in header calls).

The panel crosses PCIe in blocks on three streams (in / apply / out); the result must be the device path's, bit for bit,
whatever the blocking: several blocks, ragged last block, strided panels, odd row counts, in place, growing sizes.
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


@pytest.mark.parametrize("n,M", [(256, 12_000), (64, 45_000), (17, 300)])
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
    """A panel that cannot be page-locked for writing (read-only mapping) still goes through (pageable copies)."""
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
    Page-locking is page-granular and does not nest, so the library keeps a reference-counted registry of what it pinned;
    a call that finishes early must not unlock pages another call still copies from.  Repeated, against the device path."""
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
