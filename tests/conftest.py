import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): plain-C restatement of the reference."""
    import oracle as o

    o.build()
    return o


@pytest.fixture(scope="session")
def sri():
    """The product package; builds libsri_hip.so with hipcc if it is not there yet."""
    import experimental_gpu_programming_for_a_spectral_numerical_integration_amd as pkg

    if not os.path.exists(pkg.library_path()):
        pkg.build_library()
    pkg.load_library()
    return pkg


# ---- synthetic inputs: seed 0x5EED -------------------------------------------------
SEED = 0x5EED


def uniform_panel(rows: int, cols: int, seed: int = SEED) -> np.ndarray:
    """X_ij ~ U[-1, 1), column-major rows x cols."""
    rng = np.random.default_rng(seed)
    return np.asfortranarray(rng.uniform(-1.0, 1.0, size=(cols, rows)).T)


def rod_strain_panel(n: int, rods: int, seed: int = SEED, rows: int = None) -> np.ndarray:
    """Smooth rod-strain-field panel, the cancellation-heavy case.

    Per rod: qe ~ U[-2,2)^9, K(X_i) = Phi(X_i) qe at the Chebyshev nodes (utilities.h:49-67,
    main.cpp:69) -> N x 3, 4th column cos(|K|).  Returns column-major rows x (4*rods), using the
    first `rows` nodes (rows = n for D_N, n-1 for the integration operators).
    """
    rows = n if rows is None else rows
    rng = np.random.default_rng(seed)
    x = 0.5 * (1 + np.cos(np.pi * np.arange(n) / (n - 1)))
    t = 2 * x - 1
    P = np.stack([np.ones_like(t), t, 0.5 * (3 * t * t - 1)], axis=1)       # n x 3 Legendre basis
    qe = rng.uniform(-2.0, 2.0, size=(rods, 3, 3))                          # rod, strain comp, mode
    K = np.einsum("ne,rae->rna", P, qe)                                     # rod, node, comp
    fourth = np.cos(np.linalg.norm(K, axis=2, keepdims=True))
    panel = np.concatenate([K, fourth], axis=2)                             # rod, node, 4
    out = panel[:, :rows, :].transpose(0, 2, 1).reshape(rods * 4, rows)     # (rod, comp) x node
    return np.asfortranarray(out.T)


def parity_metrics(A: np.ndarray, X: np.ndarray, Y: np.ndarray, Yref: np.ndarray):
    """(Frobenius relative error, max backward error |E| / (|A||X|)) — BASELINE.md §4."""
    E = Y - Yref
    fro = np.linalg.norm(E) / max(np.linalg.norm(Yref), np.finfo(float).tiny)
    scale = np.abs(A) @ np.abs(X)
    scale[scale == 0] = np.finfo(float).tiny
    back = float(np.max(np.abs(E) / scale))
    return float(fro), back


FRO_TOL = 1e-12                      # north_star: <= 1e-12 relative (oracle.FRO_TOL is the same number)


def backward_tol(n: int) -> float:   # 8 N 2^-53
    return 8.0 * n * 2.0 ** -53


def assert_apply_parity(oracle, A, X, Y, n, what="", ascending=None, kind=None, which=None):
    """The parity gate of the apply — `oracle.parity_gate`, the same function bench.py exits non-zero on:

    1. bit for bit equal to the CPU model of the kernel's arithmetic (oracle/sri_mfma_model.c): every output element is
       a chain of fused multiply-adds — v_mfma_f64_16x16x4_f64 is four FMAs over its k slots — over k in the order the
       handle reports (sri_apply_summation_order): the reference's own ascending order (main.cpp:172) for differentiation
       operators of more than 512 rows, else evens before odds inside every block of 8 k;
    2. against the CPU oracle (multiply, then add: two roundings per term) <= 1e-12 in the Frobenius norm and
       <= 8 N 2^-53 in the backward error max |E| / (|A||X|);
    3. ONLY for smooth rod-strain fields (`kind == "rod"`) under the differentiation operator (`which == 0`, D_N) at
       N >= 1024 — where the oracle ITSELF is 9.4e-13 away from a binary128 product (||D||_inf = 2.1e6, |Y| ~ 1), so two
       correct fp64 results may differ by more than 1e-12 — the Frobenius bound is replaced by: GPU no farther from the
       binary128 product than the oracle is, and within 2e-12 of the oracle.  Uniform panels and the integration
       operators (measured there: 1e-16) never take that tier.
    Returns the measured figures."""
    if ascending is None:
        ascending = A.shape[0] > 512
    limited = kind == "rod" and which == 0 and n >= 1024
    out = oracle.parity_gate(A, X, Y, n, ascending=ascending, cancellation_limited=limited)
    assert out["ok"], (what, out["failures"], out)
    return out
