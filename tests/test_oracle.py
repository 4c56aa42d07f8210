"""Pins the CPU oracle: known answers from the reference's documentation, an independent
60-digit evaluation, structural identities, and the SURVEY Appendix-B anchors for config 1.
(The reference ships no tests or vectors and cannot be built here: "parity unpinned".)"""
import json
import os

import numpy as np
import pytest

from conftest import uniform_panel

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_points_descend_from_L_to_0(oracle):
    # chebyshev_differentiation.h:26: x_0 = L (tip), x_{N-1} = 0 (base)
    for n in (2, 3, 16, 64, 256, 1024):
        x = oracle.points(n)
        assert x[0] == 1.0 and abs(x[-1]) < 1e-16
        assert np.all(np.diff(x) < 0)
        assert np.allclose(x + x[::-1], 1.0, atol=2e-16)
    assert np.array_equal(oracle.points(16, 2.0), 2.0 * oracle.points(16, 1.0))


def test_coeffs(oracle):
    # chebyshev_differentiation.h:45,48
    c = oracle.coeffs(6)
    assert c.tolist() == [2.0, -1.0, 1.0, -1.0, 1.0, -2.0]
    assert oracle.coeffs(2).tolist() == [2.0, -2.0]


def test_dn_known_answers(oracle):
    # rod_modeling.pdf eq. B.7 and B.12 are on [-1,1]; the reference's domain is [0,1] => x2
    assert np.array_equal(oracle.dn(2), np.array([[1.0, -1.0], [1.0, -1.0]]))
    assert np.allclose(oracle.dn(3), np.array([[3, -4, 1], [1, 0, -1], [-1, 4, -3]], dtype=float), atol=1e-15)


def test_dn_against_the_matrices_printed_in_the_reference_documentation(oracle):
    """tests/golden/pdf_appendix_b.json: eq. B.7 / B.12 of /root/reference/materials/rod_modeling.pdf as printed — the only
    numerical values the reference tree itself holds for this path (a reference-held fixture, on [-1, 1])."""
    pdf = json.load(open(os.path.join(GOLD, "pdf_appendix_b.json")))
    scale = pdf["scale_to_reference_domain_0_1"]
    assert np.array_equal(oracle.dn(2), scale * np.array(pdf["eq_B7_D_two_nodes"]))
    assert np.allclose(oracle.dn(3), scale * np.array(pdf["eq_B12_D_three_nodes"]), rtol=0, atol=1e-15)


@pytest.mark.parametrize("n", [16, 64, 256, 1024])
def test_dn_corner_and_offdiag_formulas(oracle, n):
    D = oracle.dn(n)
    m = n - 1
    corner = (2 * m * m + 1) / 3.0                       # eq. B.14 scaled to [0,1]
    assert abs(D[0, 0] - corner) <= 1e-11 * corner
    assert abs(D[m, m] + corner) <= 1e-11 * corner
    x, c = oracle.points(n), oracle.coeffs(n)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    off = i != j
    expect = (c[i[off]] / c[j[off]]) / (x[i[off]] - x[j[off]])   # eq. B.16, same fp64 ops
    assert np.array_equal(D[off], expect)
    assert abs(np.abs(D).sum(axis=1).max() - 2.0 * m * m) <= 1e-9 * 2 * m * m   # ||D||_inf = 2(N-1)^2


@pytest.mark.parametrize("n", [2, 3, 5, 8, 16, 33, 64])
def test_dn_against_independent_60_digit_evaluation(oracle, n):
    g = np.load(os.path.join(GOLD, "dn_exact_mp.npz"))
    D, Dex = oracle.dn(n), g[f"D{n}"]
    # fp64 forming error, normwise: 2.8e-15 (N=16), 1.2e-13 (N=64) — SURVEY Appendix B
    assert np.linalg.norm(D - Dex) / np.linalg.norm(Dex) <= 2e-13
    assert np.allclose(oracle.points(n), g[f"x{n}"], atol=2e-16)
    if n >= 3:
        Dinv, _ = oracle.integration(n)
        assert np.linalg.norm(Dinv - g[f"Dinv{n}"]) / np.linalg.norm(g[f"Dinv{n}"]) <= 1e-12


def test_large_grids_against_independent_60_digit_evaluation(oracle):
    """The sizes the metric is quoted on.  tests/golden/dn_exact_mp_large.npz (make_golden.py --large): exact-maths D_N and
    Dn_NN^-1 at N=256 in full, and at N=1024 sixteen rows + the diagonal of D_N and twelve columns + rows of Dn_NN^-1
    (iterative refinement in 60-digit arithmetic).  Gates = 2x what this host measures; SURVEY Appendix B predicts
    7.2e-13 for the normwise forming error of D_256 (node rounding: one ulp in x moves near-diagonal entries by 5.9e-12)."""
    g = np.load(os.path.join(GOLD, "dn_exact_mp_large.npz"))
    fro = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert np.abs(oracle.points(256) - g["x256"]).max() <= 4e-16 and np.abs(oracle.points(1024) - g["x1024"]).max() <= 4e-16
    assert fro(oracle.dn(256), g["D256"]) <= 1.5e-12                  # measured 7.18e-13
    assert fro(oracle.integration(256)[0], g["Dinv256"]) <= 1e-13     # measured 2.1e-14
    D = oracle.dn(1024)
    assert fro(D[g["D1024_row_idx"]], g["D1024_rows"]) <= 2e-11       # measured 6.0e-12
    assert fro(np.diag(D), g["D1024_diag"]) <= 3e-11                  # measured 7.4e-12
    Dinv = oracle.integration(1024)[0]
    idx = g["Dinv1024_idx"]
    assert fro(Dinv[:, idx], g["Dinv1024_cols"]) <= 1e-12             # measured 1.9e-13
    assert fro(Dinv[idx, :], g["Dinv1024_rows"]) <= 1e-12             # measured 1.0e-13


def test_integration_matrix_1024_extended_precision_residual(oracle):
    """Whole-matrix check at N=1024: Dn_NN . Dn_NN^-1 - I evaluated in 80-bit arithmetic.  Because ||Dn_NN^-1||_inf = 1
    (checked below), the largest residual entry of a column bounds the forward error of every entry of that column."""
    n = 1024
    Dinv, _ = oracle.integration(n)
    Dnn = oracle.dn(n)[:-1, :-1]
    L = np.longdouble
    R = Dnn.astype(L) @ Dinv.astype(L) - np.eye(n - 1, dtype=L)
    assert float(np.abs(R).max()) <= 1e-11                            # measured 2.6e-13; cond_2 = 8.3e5
    assert abs(np.abs(Dinv).sum(axis=1).max() - 1.0) <= 1e-11


@pytest.mark.parametrize("n", [16, 64, 256])
def test_dn_structural_identities(oracle, n):
    D = oracle.dn(n)
    x = oracle.points(n)
    norm = 2.0 * (n - 1) ** 2
    assert np.abs(D.sum(axis=1)).max() <= 64 * n * 2.0 ** -53 * norm       # rows annihilate constants (h:104)
    assert np.abs(D @ x - 1.0).max() <= 64 * n * 2.0 ** -53 * norm        # D x = 1
    assert np.abs(D @ (x ** 3) - 3 * x ** 2).max() <= 64 * n * 2.0 ** -53 * norm


@pytest.mark.parametrize("n", [3, 16, 64, 256])
def test_integration_matrix(oracle, n):
    Dinv, Din = oracle.integration(n)
    D = oracle.dn(n)
    Dnn, Din2 = oracle.blocks(D)
    assert np.array_equal(Din, Din2) and np.array_equal(Dnn, D[:-1, :-1]) and np.array_equal(Din, D[:-1, -1])
    I = np.eye(n - 1)
    cond = {3: 10, 16: 186, 64: 3.18e3, 256: 5.17e4}[n]
    assert np.abs(Dinv @ Dnn - I).max() <= 64 * cond * 2.0 ** -53
    assert abs(np.abs(Dinv).sum(axis=1).max() - 1.0) <= 1e-12              # ||Dn_NN^-1||_inf = 1
    # spectral integration of f' = 1 from the base (x=0, value 0) reproduces f = x at the unknown nodes
    x = oracle.points(n)[:-1]
    assert np.abs(Dinv @ np.ones(n - 1) - x).max() <= 64 * n * 2.0 ** -53


def test_inverse_matches_numpy(oracle):
    rng = np.random.default_rng(1)
    for m in (1, 2, 5, 17, 60):
        A = rng.normal(size=(m, m)) + m * np.eye(m)
        assert np.allclose(oracle.inverse(A), np.linalg.inv(A), rtol=1e-11, atol=1e-13)
    with pytest.raises(RuntimeError):
        oracle.inverse(np.zeros((3, 3)))


def test_ivp_rank1(oracle):
    Din = np.array([1.0, -2.0, 3.0])
    assert np.array_equal(oracle.ivp(Din, np.array([2.0, 0.0, -1.0])), np.outer(Din, [2.0, 0.0, -1.0]))


def test_apply_forms_are_bit_identical(oracle):
    """axpy-ordered, dot-ordered and multi-threaded oracle applies give the same bits."""
    for n, M in [(16, 7), (64, 130), (255, 33)]:
        A = uniform_panel(n, n, seed=3)
        X = uniform_panel(n, M, seed=4)
        Y = oracle.apply(A, X)
        assert np.array_equal(Y, oracle.apply(A, X, dot=True))
        assert np.array_equal(Y, oracle.apply(A, X, threads=0))
        assert np.allclose(Y, A @ X, rtol=0, atol=1e-12)


def test_integrate_is_apply_of_shifted_rhs(oracle):
    n, M = 16, 9
    Dinv, Din = oracle.integration(n)
    B = uniform_panel(n - 1, M, seed=8)
    xb = np.linspace(-1, 1, M)
    assert np.array_equal(oracle.integrate(Dinv, Din, B, xb), oracle.apply(Dinv, np.asfortranarray(B - np.outer(Din, xb))))


def test_legendre_basis_and_strain(oracle):
    # utilities.h:54-63: x = 2X-1, P0=1, P1=x, P2=(3x^2-1)/2
    for X in (0.0, 0.25, 0.7, 1.0):
        t = 2 * X - 1
        assert np.allclose(oracle.legendre_basis(X, 3), [1.0, t, 0.5 * (3 * t * t - 1)], atol=1e-16)
    qe = np.arange(1.0, 10.0)
    t = 2 * 0.3 - 1
    P = np.array([1.0, t, 0.5 * (3 * t * t - 1)])
    assert np.allclose(oracle.strain(0.3, qe), [P @ qe[0:3], P @ qe[3:6], P @ qe[6:9]], atol=1e-15)


def test_position_rhs_is_first_rotation_column(oracle):
    # main.cpp:136: q.toRotationMatrix() * e1, no normalisation
    rng = np.random.default_rng(2)
    Q = np.asfortranarray(rng.normal(size=(5, 4)))
    b = oracle.position_rhs(Q)
    w, x, y, z = Q.T
    ref = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)], axis=1)
    assert np.allclose(b, ref, atol=1e-15)


def test_config1_main_cpp_against_survey_anchors(oracle):
    """BASELINE config 1: N=16, one rod, qe of main.cpp:187-195 — Q_stack and r_stack."""
    anchors = json.load(open(os.path.join(GOLD, "config1_anchors.json")))
    n = anchors["n"]
    Q = oracle.integrate_quaternions(n, np.array(anchors["qe"]))
    r = oracle.integrate_position(n, Q)
    assert np.array_equal(np.array(anchors["qe"]), oracle.QE_MAIN_CPP)
    for node in anchors["nodes"]:
        i = node["i"]
        assert abs(Q[i, 0] - node["q_w"]) <= 5e-15 and abs(Q[i, 2] - node["q_y"]) <= 5e-15
        assert abs(Q[i, 1]) <= 1e-15 and abs(Q[i, 3]) <= 1e-15
        assert abs(r[i, 0] - node["r_x"]) <= 5e-15 and abs(r[i, 2] - node["r_z"]) <= 5e-15 and abs(r[i, 1]) <= 1e-15
    # physics: planar bend, tip angle theta(1) = qe[3]; spectral result agrees to ~2e-11 (discretisation)
    tip = anchors["analytic_tip"]
    assert np.abs(Q[0] - np.array(tip["q"])).max() <= 1e-10
    assert np.abs(r[0] - np.array(tip["r"])).max() <= 1e-10
    assert np.abs(np.linalg.norm(Q, axis=1) - 1).max() <= 1e-11


def test_error_codes(oracle):
    import ctypes as C
    lib = oracle.lib()
    buf = (C.c_double * 4)()
    assert lib.sri_oracle_points(1, 1.0, buf) == -1
    assert lib.sri_oracle_dn(4, buf, 3) == -1


def test_mfma_arithmetic_models(oracle):
    """oracle/sri_mfma_model.c: the CPU models of the kernel's summation.  MODEL_ASCENDING_MULADD is the oracle's own
    arithmetic bit for bit; the FMA models differ from it by rounding only; exact small-integer products agree in every
    model; the binary128 yardstick is at least as close to every model as they are to each other."""
    from conftest import rod_strain_panel

    rng = np.random.default_rng(5)
    for rows, M in [(16, 9), (100, 33), (255, 17), (256, 40)]:
        A = uniform_panel(rows, rows, seed=rows)
        X = uniform_panel(rows, M, seed=rows + 1)
        Y = oracle.apply(A, X)
        assert np.array_equal(oracle.model_apply(A, X, oracle.MODEL_ASCENDING_MULADD), Y)
        Yx = oracle.apply_exact(A, X)
        for mode in range(6):
            Ym = oracle.model_apply(A, X, mode)
            assert np.linalg.norm(Ym - Yx) <= 4 * rows * 2.0 ** -53 * np.linalg.norm(np.abs(A) @ np.abs(X))
        Ai = np.asfortranarray(rng.integers(-7, 8, size=(rows, rows)).astype(np.float64))
        Xi = np.asfortranarray(rng.integers(-9, 10, size=(rows, M)).astype(np.float64))
        for mode in range(6):
            assert np.array_equal(oracle.model_apply(Ai, Xi, mode), Ai @ Xi)
        assert np.array_equal(oracle.apply_exact(Ai, Xi), Ai @ Xi)
    # the regime that decides the k order: smooth field under D_1024.  Equal-parity k's summed together (the 16-byte
    # fragment order) is ~1.4x farther from the true product than the reference's arithmetic; ascending k with fused
    # multiply-adds is closer than the reference itself.
    n = 1024
    A = oracle.dn(n)
    X = rod_strain_panel(n, 16)
    Yx = oracle.apply_exact(A, X)
    err = {m: np.linalg.norm(oracle.model_apply(A, X, m) - Yx) / np.linalg.norm(Yx) for m in (0, 4, 5)}
    assert err[oracle.MODEL_ASCENDING_FMA] < err[oracle.MODEL_ASCENDING_MULADD] < err[oracle.MODEL_KERNEL]
    assert err[oracle.MODEL_ASCENDING_MULADD] <= 1.2e-12 and err[oracle.MODEL_ASCENDING_FMA] <= 0.8e-12


# ---- the parity gate itself (oracle.parity_gate: what the GPU tests, smoke() and bench.py all call) --------------------
def test_parity_gate_accepts_the_kernels_arithmetic_and_rejects_anything_else(oracle):
    from conftest import rod_strain_panel, uniform_panel

    n = 64
    A = oracle.dn(n)
    X = uniform_panel(n, 96)
    Ymodel = oracle.model_apply(A, X, oracle.MODEL_KERNEL)           # what the GPU must return for a paired-order handle
    g = oracle.parity_gate(A, X, Ymodel, n, ascending=False)
    assert g["ok"] and g["gate"] == oracle.GATE_FRO and g["bit_exact_vs_fma_model"] and g["failures"] == []
    assert g["fro_vs_oracle"] <= 1e-12 and g["backward"] <= g["backward_bound"]
    # the same numbers judged against the OTHER k order: close to the oracle, but not the kernel's bits
    g = oracle.parity_gate(A, X, Ymodel, n, ascending=True)
    assert not g["ok"] and not g["bit_exact_vs_fma_model"] and "FMA-chain model" in g["failures"][0]
    # one element off by one ulp: every norm still passes, bit-equality does not
    Y1 = Ymodel.copy()
    Y1[3, 5] = np.nextafter(Y1[3, 5], np.inf)
    g = oracle.parity_gate(A, X, Y1, n, ascending=False)
    assert not g["ok"] and g["fro_vs_oracle"] <= 1e-12
    # a wrong operator entry (the `make broken` build of the library: 2^-20 relative): caught by the norms as well
    A2 = A.copy()
    A2[0, 0] *= 1 + 2.0 ** -20
    g = oracle.parity_gate(A, X, oracle.model_apply(A2, X, oracle.MODEL_KERNEL), n, ascending=False)
    assert not g["ok"] and any("fro" in f for f in g["failures"])
    # NaN in the result
    Y2 = Ymodel.copy()
    Y2[0, 0] = np.nan
    assert not oracle.parity_gate(A, X, Y2, n, ascending=False)["ok"]


def test_parity_gate_cancellation_tier_is_only_for_the_declared_regime(oracle):
    """A product that is 1.5e-12 from the oracle (but closer to binary128 than the oracle is) passes ONLY when the caller
    declares the cancellation-limited regime; and not even then without the binary128 product."""
    from conftest import rod_strain_panel

    n = 1024                                                         # the one regime where the oracle itself is ~0.9e-12 from exact
    A = oracle.dn(n)
    X = rod_strain_panel(n, 8)
    Yx = oracle.apply_exact(A, X)                                    # binary128 product, rounded once
    Yref = oracle.apply(A, X, threads=0)
    E = Yref - Yx
    assert 5e-13 < np.linalg.norm(E) / np.linalg.norm(Yx) < 1.5e-12
    Y = Yx - 0.3 * E                                                 # closer to exact than the oracle, on the far side of it
    fro = np.linalg.norm(Y - Yref) / np.linalg.norm(Yref)
    assert 1e-12 < fro <= 2e-12
    # (bit-equality with the model fails for this synthetic Y in every call; what is tested is the Frobenius tier)
    plain = oracle.parity_gate(A, X, Y, n, ascending=False)
    assert any(f.startswith("fro") for f in plain["failures"]) and plain["gate"] == oracle.GATE_FRO
    declared = oracle.parity_gate(A, X, Y, n, ascending=False, cancellation_limited=True)
    assert declared["gate"] == oracle.GATE_CANCELLATION and not any(f.startswith("fro") for f in declared["failures"])
    no_exact = oracle.parity_gate(A, X, Y, n, ascending=False, cancellation_limited=True, exact=False)
    assert any("binary128 product skipped" in f for f in no_exact["failures"])
