#!/usr/bin/env python3
"""panel_predictor_study.py — CPU-only numpy study for DESIGN §10 / EXPERIMENTS R5.11 (no library, no oracle, nothing timed).

Question: the rod predictor solves q' = 1/2 A(K(s)) q, q(0) = q0 by Picard sweeps on ONE 32-node Chebyshev interval (15 sweeps
behind a 16-node level).  Would P sub-intervals of 16 nodes each, solved one after the other (each started from the tip of the
one before), give a start of the same quality in fewer sweep-units?  A sweep on <= 16 nodes is one 16x16 MFMA tile row (1/4 of
a 32-node sweep).  Reported per variant, over `--rods` random rods (strain coefficients ~ U[-scale, scale)):
  start_err   max |Q_start - Q_fine| at the fine nodes (fine = collocation on N = --n nodes, dense solve)
  first_upd   max |G1 - G0| of ONE fine Picard sweep from that start (what the closing product's criterion bounds)
  sweeps      Picard sweeps per panel to max|update| <= 2e-14, and the cost in units of one 32-node sweep
Plain collocation on every grid (the product's coarse level uses the FINE operator as the coarse nodes see it — EXP "(old §7)" —
which this study does not model: read start_err / first_upd as a comparison BETWEEN variants, not as the product's numbers)."""
import argparse

import numpy as np


def cheb(n):
    """nodes on [0, 1] ascending from s = 0, and the differentiation matrix there (n nodes)."""
    j = np.arange(n)
    x = -np.cos(np.pi * j / (n - 1))                 # [-1, 1] ascending
    c = np.ones(n); c[0] = c[-1] = 2.0
    c = c * (-1.0) ** j
    X = np.tile(x, (n, 1)).T
    dX = X - X.T + np.eye(n)
    D = np.outer(c, 1.0 / c) / dX
    D = D - np.diag(D.sum(axis=1))
    return (x + 1) / 2, D * 2.0                      # d/ds on [0, 1]


def bary_matrix(xs, xt):
    """interpolation from Chebyshev-Lobatto nodes xs (any affine image) to points xt."""
    n = len(xs)
    w = (-1.0) ** np.arange(n); w[0] *= 0.5; w[-1] *= 0.5
    P = np.zeros((len(xt), n))
    for i, t in enumerate(xt):
        d = t - xs
        hit = np.where(np.abs(d) < 1e-15)[0]
        if len(hit):
            P[i, hit[0]] = 1.0
        else:
            P[i] = (w / d) / np.sum(w / d)
    return P


def legendre3(s):
    x = 2 * s - 1
    return np.stack([np.ones_like(x), x, 0.5 * (3 * x * x - 1)], axis=-1)


def A_of(K):
    k1, k2, k3 = K[..., 0], K[..., 1], K[..., 2]
    z = np.zeros_like(k1)
    return np.stack([np.stack([z, -k1, -k2, -k3], -1), np.stack([k1, z, k3, -k2], -1),
                     np.stack([k2, -k3, z, k1], -1), np.stack([k3, k2, -k1, z], -1)], -2)


def picard(s, D, coef, q0, a, b, tol=2e-14, max_it=200, Q=None):
    """q' = 1/2 A(K(a + (b - a) t)) q on t in [0, 1] scaled to [a, b]; returns Q at the nodes and the sweep count."""
    n = len(s)
    Dnn_inv = np.linalg.inv(D[1:, 1:] / (b - a))
    w = Dnn_inv @ (D[1:, 0] / (b - a))
    K = legendre3(a + (b - a) * s) @ coef.T           # (n, 3)
    Ah = 0.5 * A_of(K)                                # (n, 4, 4)
    if Q is None:
        Q = np.tile(q0, (n, 1))
    Gold = None
    for it in range(1, max_it + 1):
        G = np.einsum("nij,nj->ni", Ah, Q)
        Qn = Q.copy()
        Qn[1:] = Dnn_inv @ G[1:] - np.outer(w, q0)
        Q = Qn
        if Gold is not None and np.max(np.abs(G - Gold)) <= tol:
            return Q, it
        Gold = G
    return Q, max_it


def direct(s, D, coef, q0):
    n = len(s)
    K = legendre3(s) @ coef.T
    Ah = 0.5 * A_of(K)
    M = np.kron(D, np.eye(4))
    for i in range(n):
        M[4 * i:4 * i + 4, 4 * i:4 * i + 4] -= Ah[i]
    M[:4] = 0; M[:4, :4] = np.eye(4)
    rhs = np.zeros(4 * n); rhs[:4] = q0
    return np.linalg.solve(M, rhs).reshape(n, 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--rods", type=int, default=12)
    ap.add_argument("--scale", type=float, default=2.0)
    a = ap.parse_args()
    rng = np.random.default_rng(5)
    sf, Df = cheb(a.n)
    Dinv = np.linalg.inv(Df[1:, 1:]); wf = Dinv @ Df[1:, 0]
    q0 = np.array([1.0, 0, 0, 0])
    variants = [("1 x 32 (today)", 1, 32), ("1 x 16", 1, 16), ("2 x 16", 2, 16), ("4 x 16", 4, 16), ("2 x 32", 2, 32), ("3 x 16", 3, 16)]
    acc = {v[0]: {"start_err": 0.0, "first_upd": 0.0, "sweeps": []} for v in variants}
    for r in range(a.rods):
        coef = rng.uniform(-a.scale, a.scale, size=(3, 3))
        Qf = direct(sf, Df, coef, q0)
        Ahf = 0.5 * A_of(legendre3(sf) @ coef.T)
        for name, P, nc in variants:
            sc, Dc = cheb(nc)
            Qs = np.zeros((a.n, 4)); tip = q0; sweeps = []
            for p in range(P):
                lo, hi = p / P, (p + 1) / P
                # each panel behind its own 16-node level when it has 32 nodes (as today's predictor does)
                Qc, it = picard(sc, Dc, coef, tip, lo, hi)
                sweeps.append(it)
                m = (sf >= lo - 1e-15) & ((sf < hi) if p < P - 1 else (sf <= hi + 1e-15))
                Qs[m] = bary_matrix(lo + (hi - lo) * sc, sf[m]) @ Qc
                tip = Qc[-1]
            G0 = np.einsum("nij,nj->ni", Ahf, Qs)
            Q1 = Qs.copy(); Q1[1:] = Dinv @ G0[1:] - np.outer(wf, q0)
            G1 = np.einsum("nij,nj->ni", Ahf, Q1)
            A = acc[name]
            A["start_err"] = max(A["start_err"], np.max(np.abs(Qs - Qf)))
            A["first_upd"] = max(A["first_upd"], np.max(np.abs(G1 - G0)))
            A["sweeps"].append(sweeps)
    print(f"fine grid N = {a.n}, {a.rods} rods, strain coefficients ~ U[-{a.scale}, {a.scale})")
    print(f"{'coarse grid':16s} {'start_err':>10s} {'first_upd':>10s}  sweeps per panel (mean, from q0 / the previous tip)   cost in 32-node sweeps")
    for name, P, nc in variants:
        A = acc[name]
        sw = np.mean(np.array(A["sweeps"]), axis=0)
        unit = 0.25 if nc <= 16 else 1.0
        print(f"{name:16s} {A['start_err']:10.1e} {A['first_upd']:10.1e}  {np.round(sw, 1)}   {unit * sw.sum():.1f}")


if __name__ == "__main__":
    main()
