#!/usr/bin/env python3
"""mixing_predictor_study.py — CPU-only numpy study for EXPERIMENTS R5.12 / DESIGN §10 (no library, no oracle, nothing timed).

Question: the rod predictor's cost is the SWEEP COUNT of a Picard iteration Q <- F(Q) = Dn_NN^-1 (1/2 A(K) Q) - w q0^T on one
interval (15 sweeps on 32 nodes behind ~20 on 16).  F is affine in Q and its linear part is a discretised Volterra operator
(spectrum clustered at 0), so mixing the last iterates (Anderson acceleration of depth d, per rod: a d x d least-squares problem
over the rod's 4 (n - 1) values) is a Krylov method on it.  Same fixed point, same stopping rule (max |update of G| <= tol on the
PLAIN sweep's residual).  Reported: sweeps (= products with the operator) to the tolerance, from Q = q0, and the error against
the dense collocation solve on the same grid — per depth, over `--rods` random rods."""
import argparse

import numpy as np

from panel_predictor_study import A_of, cheb, direct, legendre3


def solve(n, coef, q0, depth, tol=2e-14, max_it=200, Qstart=None):
    s, D = cheb(n)
    Dinv = np.linalg.inv(D[1:, 1:])
    w = Dinv @ D[1:, 0]
    Ah = 0.5 * A_of(legendre3(s) @ coef.T)

    def F(Qu):                                        # Qu: unknown nodes (n-1, 4)
        Q = np.vstack([q0[None, :], Qu])
        G = np.einsum("nij,nj->ni", Ah, Q)
        return Dinv @ G[1:] - np.outer(w, q0)

    x = np.tile(q0, (n - 1, 1)) if Qstart is None else Qstart.copy()
    X, R = [], []
    for it in range(1, max_it + 1):
        fx = F(x)
        r = fx - x
        if np.max(np.abs(r)) <= tol:
            return np.vstack([q0[None, :], fx]), it
        X.append(x.ravel().copy()); R.append(r.ravel().copy())
        X, R = X[-(depth + 1):], R[-(depth + 1):]
        if depth == 0 or len(R) < 2:
            x = fx
            continue
        dR = np.stack([R[i + 1] - R[i] for i in range(len(R) - 1)], axis=1)
        dX = np.stack([X[i + 1] - X[i] for i in range(len(X) - 1)], axis=1)
        g, *_ = np.linalg.lstsq(dR, R[-1], rcond=None)
        x = (X[-1] + R[-1] - (dX + dR) @ g).reshape(n - 1, 4)
    return np.vstack([q0[None, :], F(x)]), max_it


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rods", type=int, default=24)
    ap.add_argument("--scale", type=float, default=2.0)
    a = ap.parse_args()
    rng = np.random.default_rng(5)
    q0 = np.array([1.0, 0, 0, 0])
    coefs = [rng.uniform(-a.scale, a.scale, size=(3, 3)) for _ in range(a.rods)]
    print(f"qe ~ U[-{a.scale}, {a.scale}), {a.rods} rods, tol 2e-14 on the plain sweep's update")
    print(f"{'grid':>6} {'depth':>5} {'sweeps mean':>12} {'max':>4} {'max err vs dense solve':>24}")
    for n in (16, 32):
        s, D = cheb(n)
        ref = [direct(s, D, c, q0) for c in coefs]
        for depth in (0, 1, 2, 3, 4):
            its, err = [], 0.0
            for c, Qd in zip(coefs, ref):
                Q, it = solve(n, c, q0, depth)
                its.append(it); err = max(err, np.max(np.abs(Q - Qd)))
            print(f"{n:>6} {depth:>5} {np.mean(its):>12.1f} {max(its):>4} {err:>24.2e}")
    # the product's nesting: the 32-node level starts from the interpolated 16-node solution
    from panel_predictor_study import bary_matrix
    s16, _ = cheb(16); s32, _ = cheb(32)
    P = bary_matrix(s16, s32)
    print("32 nodes started from the 16-node solution (the product's nesting):")
    for depth in (0, 1, 2, 3):
        its16, its32 = [], []
        for c in coefs:
            Q16, i16 = solve(16, c, q0, depth)
            Q32, i32 = solve(32, c, q0, depth, Qstart=(P @ Q16)[1:])
            its16.append(i16); its32.append(i32)
        print(f"  depth {depth}: 16-node sweeps {np.mean(its16):.1f}, 32-node sweeps {np.mean(its32):.1f}, "
              f"cost in 32-node sweep units {np.mean(its16) / 4 + np.mean(its32):.1f}")


if __name__ == "__main__":
    main()
