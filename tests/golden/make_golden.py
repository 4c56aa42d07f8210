#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

Nothing here comes from the reference binary: it cannot be built in this image (Eigen/Boost
absent) and it ships no tests or vectors ("parity unpinned").  The fixtures are:

  dn_exact_mp.npz      D_N for N in {2,3,5,8,16,33,64} evaluated INDEPENDENTLY of the oracle: mpmath
                       at 60 digits from the closed form of the reference's documentation
                       (materials/rod_modeling.pdf eq. B.1, B.14-B.16, scaled to [0,1]), rounded to
                       fp64 at the end.  Also the 60-digit inverse of the (N-1)x(N-1) block.
  config1_anchors.json the SURVEY.md Appendix B table for BASELINE config 1 (N=16, one rod, the
                       hard-coded qe of main.cpp:187-195), copied from SURVEY.md — survey-computed
                       (cross-checked there against a 40-digit solve and the analytic planar bend).
  dn_oracle_n64.npz    getDn<64>, Dn_NN^-1 and the nodes x[] as the ORACLE computes them on this
                       host (regression pin for the C++ host formers; N <= 64 is insensitive to a
                       1-ulp libm difference: 1.8e-13).

  dn_exact_mp_large.npz the same independent 60-digit evaluation at the sizes the metric is quoted on
                       (`--large`, minutes):  N=256: full D, full Dn_NN^-1 (mp LU), x[];  N=1024: x[], 16 full
                       rows of D, its diagonal, and 12 columns + 12 rows of the exact Dn_NN^-1 obtained by
                       iterative refinement in 60-digit arithmetic (residual against the exact-math Dn_NN in mp,
                       correction with an fp64 inverse; contraction ~1e-10 per sweep, 5 sweeps).
  pdf_appendix_b.json  the two matrices PRINTED in the reference's own documentation
                       (/root/reference/materials/rod_modeling.pdf, eq. B.7 and B.12, on [-1, 1]) — the only
                       numerical values the reference tree holds for this path, typed in from the PDF.

  config1_stdout.txt   stdout of the reference's own, UNMODIFIED /root/reference/main.cpp compiled in place against
                       this repo's drop-in header, libsri_hip.so and the stand-in headers under compat/
                       (`--stdout`; tests/test_reference_main.py rebuilds it and compares).  Product output, checked
                       there against config1_anchors.json and the oracle — not a pin of the oracle.

Run:  python tests/golden/make_golden.py           (needs mpmath; takes a few seconds)
      python tests/golden/make_golden.py --stdout  (config1_stdout.txt; needs /root/reference and a built libsri_hip.so)
      python tests/golden/make_golden.py --large   (additionally dn_exact_mp_large.npz; ~10 min on 8 cores)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def dn_exact(n: int, digits: int = 60):
    import mpmath as mp

    mp.mp.dps = digits
    x = [(1 + mp.cos(mp.pi * j / (n - 1))) / 2 for j in range(n)]          # eq. B.1 on [0,1]
    c = [(2 if j in (0, n - 1) else 1) * (-1) ** j for j in range(n)]
    D = mp.zeros(n, n)
    for i in range(n):
        for j in range(n):
            if i != j:
                D[i, j] = mp.mpf(c[i]) / c[j] / (x[i] - x[j])                # eq. B.16
    for i in range(n):
        D[i, i] = -sum(D[i, j] for j in range(n) if j != i)                  # rows annihilate constants
    Dnn = D[0:n - 1, 0:n - 1]
    Dinv = Dnn ** -1
    to_np = lambda Mx: np.array([[float(Mx[i, j]) for j in range(Mx.cols)] for i in range(Mx.rows)])
    return to_np(D), to_np(Dinv), np.array([float(v) for v in x])


def _mp_setup(n, digits=60):
    import mpmath as mp

    mp.mp.dps = digits
    x = [(1 + mp.cos(mp.pi * j / (n - 1))) / 2 for j in range(n)]
    c = [(2 if j in (0, n - 1) else 1) * (-1) ** j for j in range(n)]
    return mp, x, c


def _mp_row(mp, x, c, i):
    """Row i of the exact D_N (eq. B.16 off the diagonal, negative row sum on it) as a list of mpf."""
    n = len(x)
    row = [mp.mpf(0)] * n
    for j in range(n):
        if j != i:
            row[j] = mp.mpf(c[i]) / c[j] / (x[i] - x[j])
    row[i] = -mp.fsum(row)
    return row


def _refine_worker(args):
    """Exact column (kind 'c') or row (kind 'r') `idx` of Dn_NN^-1 at N = n by iterative refinement in mp."""
    n, kind, idx, z0, Dinv64 = args
    mp, x, c = _mp_setup(n)
    m = n - 1
    rows = [_mp_row(mp, x, c, i)[:m] for i in range(m)]                      # exact-math Dn_NN
    if kind == "r":                                                          # rows of the inverse: solve Dnn^T y = e_idx
        rows = [[rows[i][j] for i in range(m)] for j in range(m)]
        Dinv64 = Dinv64.T
    z = [mp.mpf(float(v)) for v in z0]
    for _ in range(5):
        res = [(1 if i == idx else 0) - mp.fdot(rows[i], z) for i in range(m)]
        r64 = np.array([float(v) for v in res])
        d = Dinv64 @ r64
        z = [z[i] + mp.mpf(float(d[i])) for i in range(m)]
    res = max(abs((1 if i == idx else 0) - mp.fdot(rows[i], z)) for i in range(m))
    return kind, idx, np.array([float(v) for v in z]), float(res)


def large():
    """dn_exact_mp_large.npz (see the module docstring)."""
    import multiprocessing as mpc

    import oracle

    out = {}
    D, Dinv, x = dn_exact(256)
    out["D256"], out["Dinv256"], out["x256"] = D, Dinv, x
    n = 1024
    mp, xs, c = _mp_setup(n)
    out["x1024"] = np.array([float(v) for v in xs])
    ridx = np.array([0, 1, 2, 3, 100, 255, 256, 511, 512, 700, 767, 1000, 1020, 1021, 1022, 1023])
    out["D1024_row_idx"] = ridx
    out["D1024_rows"] = np.array([[float(v) for v in _mp_row(mp, xs, c, int(i))] for i in ridx])
    diag = []
    for i in range(n):
        diag.append(float(-mp.fsum(mp.mpf(c[i]) / c[j] / (xs[i] - xs[j]) for j in range(n) if j != i)))
    out["D1024_diag"] = np.array(diag)
    Dinv64, _ = oracle.integration(n)                                        # only the preconditioner / start of the refinement
    cidx = np.array([0, 1, 2, 50, 255, 511, 512, 800, 1000, 1020, 1021, 1022])
    jobs = [(n, "c", int(j), Dinv64[:, j].copy(), np.ascontiguousarray(Dinv64)) for j in cidx]
    jobs += [(n, "r", int(i), Dinv64[i, :].copy(), np.ascontiguousarray(Dinv64)) for i in cidx]
    with mpc.get_context("spawn").Pool(8) as pool:
        res = pool.map(_refine_worker, jobs)
    cols = {j: z for k, j, z, _ in res if k == "c"}
    rws = {i: z for k, i, z, _ in res if k == "r"}
    worst = max(r for _, _, _, r in res)
    assert worst < 1e-40, worst                                              # the refinement converged to working precision
    out["Dinv1024_idx"] = cidx
    out["Dinv1024_cols"] = np.stack([cols[int(j)] for j in cidx], axis=1)
    out["Dinv1024_rows"] = np.stack([rws[int(i)] for i in cidx], axis=0)
    np.savez_compressed(os.path.join(HERE, "dn_exact_mp_large.npz"), **out)
    print("dn_exact_mp_large.npz written; refinement residual", worst)


def main():
    import oracle

    if "--large" in sys.argv:
        large()
        return
    if "--stdout" in sys.argv:
        import subprocess
        import tempfile

        root = os.path.dirname(os.path.dirname(HERE))
        sys.path.insert(0, os.path.join(root, "tests"))
        from test_reference_main import build_reference_main

        with tempfile.TemporaryDirectory() as tmp:
            exe = build_reference_main(tmp)
            text = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
        open(os.path.join(HERE, "config1_stdout.txt"), "w").write(text)
        print("config1_stdout.txt written")
        return

    out = {}
    for n in (2, 3, 5, 8, 16, 33, 64):
        D, Dinv, x = dn_exact(n)
        out[f"D{n}"], out[f"Dinv{n}"], out[f"x{n}"] = D, Dinv, x
    np.savez_compressed(os.path.join(HERE, "dn_exact_mp.npz"), **out)

    n = 64
    Dinv, Din = oracle.integration(n)
    np.savez_compressed(os.path.join(HERE, "dn_oracle_n64.npz"), D=oracle.dn(n), Dinv=Dinv, Din=Din, x=oracle.points(n))

    # SURVEY.md Appendix B, "Predicted config-1 output" table (node, q_w, q_y, r_x, r_z); x,z of q and y of r are 0
    rows = [
        (0, 0.7997698672530426, 0.600306721128658, 0.56267255751617817, -0.74591365968166945),
        (1, 0.8000674379656993, 0.5999100721818162, 0.55961613596161675, -0.73542365892290473),
        (2, 0.8011019782660156, 0.598527877727065, 0.55051348914618969, -0.70443172435801382),
        (3, 0.8033723210878949, 0.5954770471714576, 0.53551619657592842, -0.65436558839914738),
        (4, 0.8078859386738393, 0.5893388754284609, 0.51470303062109157, -0.58759158199448691),
        (5, 0.8162079148316977, 0.5777582883561831, 0.48781865503506378, -0.50741654638032163),
        (6, 0.8301876348211522, 0.5574840724077008, 0.45402368892598627, -0.41811545706633307),
        (7, 0.8512341331937074, 0.5247860997436278, 0.41191423957520668, -0.32497354294414077),
        (8, 0.8792718355717443, 0.47632031152204, 0.36010448530839295, -0.23425125421014045),
        (9, 0.9118838815505612, 0.4104482751414932, 0.29844143588806021, -0.15280963229881236),
        (10, 0.9443958067410321, 0.3288108273882804, 0.22939950895262590, -0.087019217884740943),
        (11, 0.9713838654977799, 0.23751502236591632, 0.15866532216893975, -0.040906630922787693),
        (12, 0.9891400413829168, 0.14697611550898018, 0.094051304924729656, -0.014388639985179570),
        (13, 0.9975530315127384, 0.06991387072236478, 0.043083577784342332, -0.0030585591995025913),
        (14, 0.9998344166064083, 0.01819723511416487, 0.010923775864312459, -0.00019945374854395297),
    ]
    json.dump({
        "source": "SURVEY.md Appendix B (survey-computed; not output of the reference binary)",
        "n": 16,
        "qe": [0, 0, 0, 1.2877691307032, -1.63807499160786, 0.437406679142598, 0, 0, 0],
        "analytic_tip": {"theta": 1.2877691307032, "q": [0.79976986725508, 0, 0.60030672112746, 0],
                         "r": [0.56267255750165, 0, -0.74591365970371]},
        "nodes": [{"i": r[0], "q_w": r[1], "q_y": r[2], "r_x": r[3], "r_z": r[4]} for r in rows],
    }, open(os.path.join(HERE, "config1_anchors.json"), "w"), indent=1)
    # materials/rod_modeling.pdf Appendix B as printed (domain [-1, 1]); the reference's domain is [0, 1] => x2
    json.dump({
        "source": "/root/reference/materials/rod_modeling.pdf, Appendix B — values printed in the document",
        "domain": [-1, 1],
        "eq_B7_D_two_nodes": [[0.5, -0.5], [0.5, -0.5]],
        "eq_B12_D_three_nodes": [[1.5, -2.0, 0.5], [0.5, 0.0, -0.5], [-0.5, 2.0, -1.5]],
        "eq_B14_corner": "d_00 = (2 n^2 + 1)/6 = -d_nn, n = N - 1",
        "eq_B16_offdiag": "d_ij = (c_i/c_j) (-1)^(i+j) / (x_i - x_j)",
        "scale_to_reference_domain_0_1": 2.0,
    }, open(os.path.join(HERE, "pdf_appendix_b.json"), "w"), indent=1)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
