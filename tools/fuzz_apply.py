#!/usr/bin/env python3
"""fuzz_apply.py — randomised differential run of the apply against the CPU model of its arithmetic (GPU box).

    python tools/fuzz_apply.py [--cases 400] [--seed 1]

Every case draws an operator (one of the three Chebyshev operators of a random grid, or a random dense matrix), a column
count, leading dimensions, an 8-byte-aligned offset into a larger buffer and poisons everything the call must not read
into the result (NaN padding rows) or write (sentinel padding).  Gate: the result equals oracle.model_apply — the FMA
chain in the handle's k order — BIT FOR BIT, the padding is untouched, the tip row equals row 0, and the host-panel entry
point returns the same bits.  The oracle is the checker here (tools/ and tests/ may use it); nothing is timed."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import experimental_gpu_programming_for_a_spectral_numerical_integration_amd as sri  # noqa: E402
import oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    oracle.build()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda", 0)
    lib = sri.load_library()
    bad = 0
    kernels = {}
    # breadcrumbs on stderr, one short line per GPU phase, flushed BEFORE the call: a GPU fault aborts the process without a
    # Python traceback, and tests/test_fuzz_gpu.py shows the tail of stderr — the last line then names the case and the call
    def crumb(text):
        print(text, file=sys.stderr, flush=True)

    # plain pageable transfers, as any caller's would be (round 3 routed these through pinned staging while the library
    # page-locked caller memory per call; the library no longer does — csrc/sri_host_panels.hip — and the harness no longer hides)
    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def to_host(t):
        return t.cpu().numpy()

    for case in range(a.cases):
        kind = rng.choice(["cheb", "cheb", "dense"])
        if kind == "cheb":
            n = int(rng.choice([int(rng.integers(2, 70)), int(rng.integers(70, 300)), int(rng.choice([256, 257, 512, 513, 600, 1024])),
                                int(rng.choice([17, 33, 49, 65, 129, 257]))], p=[0.45, 0.35, 0.1, 0.1]))
            which = int(rng.integers(0, 3))
            if n == 2 and which != sri.OP_DN:
                which = sri.OP_DN
            op = sri.SpectralApply(n, which)
            if which == sri.OP_DN:
                A = oracle.dn(n)
            else:
                Dinv, Din = oracle.integration(n)
                A = Dinv if which == sri.OP_DNN_INV else np.asfortranarray(oracle.dn(n)[: n - 1, : n - 1])
        else:
            rows = int(rng.integers(1, 300))
            A = np.asfortranarray(rng.uniform(-1, 1, size=(rows, rows)))
            op = sri.SpectralApply(rows + 1, sri.OP_DN, matrix=A)
            which = -1
        rows = op.rows
        op.set_column_plan(case % 2 == 1)                        # odd cases: whatever the column plan picks for M columns
        big = rows > 300
        M = int(rng.integers(1, 600 if big else 6000))
        if rng.random() < 0.15:
            M = int(rng.choice([1, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257]))
        if case % 2 == 1 and 129 <= rows <= 300 and rng.random() < 0.08:
            M = int(rng.integers(33000, 60000))                  # more than a round of 256 x 128 tiles: plans in two geometries
        ldx = rows + int(rng.choice([0, 0, 1, 2, 3, 7, 16]))
        ldy = rows + int(rng.choice([0, 0, 1, 2, 5, 8]))
        offx, offy = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        Xh = np.full((M, ldx), np.nan)
        Xh[:, :rows] = rng.uniform(-1, 1, size=(M, rows))
        if rng.random() < 0.3:                                   # smooth columns: the cancellation-heavy kind
            t = np.linspace(-1, 1, rows)
            Xh[:, :rows] = np.cos(np.outer(rng.uniform(0.2, 3, M), t) + rng.uniform(0, 6, (M, 1)))
        u = rng.random()
        if u < 0.08:
            Xh[:, :rows] *= 2.0 ** -1015                         # results and partial sums in the subnormal range: no flush to zero
        elif u < 0.12:
            Xh[:, :rows] *= 2.0 ** 900                           # large but finite
        xbuf = torch.full((offx + M * ldx + 8,), float("nan"), dtype=torch.float64, device=dev)
        xbuf[offx:offx + M * ldx] = to_dev(Xh.reshape(-1))
        ybuf = torch.full((offy + M * ldy + 8,), -7.25, dtype=torch.float64, device=dev)
        tips = torch.full((M,), 3.5, dtype=torch.float64, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        xp, yp = xbuf.data_ptr() + 8 * offx, ybuf.data_ptr() + 8 * offy
        use_tips = rng.random() < 0.5
        crumb(f"case {case}: {kind} rows={rows} which={which} M={M} ldx={ldx} ldy={ldy} off=({offx},{offy}) tips={use_tips}: apply")
        if use_tips:
            rc = lib.sri_apply_tips(op._h, xp, ldx, yp, ldy, M, tips.data_ptr(), s)
        else:
            rc = lib.sri_apply(op._h, xp, ldx, yp, ldy, M, s)
        torch.cuda.synchronize()
        what = f"case {case}: {kind} rows={rows} which={which} M={M} ldx={ldx} ldy={ldy} off=({offx},{offy}) tips={use_tips} {op.last_kernel}"
        kernels[op.last_kernel] = kernels.get(op.last_kernel, 0) + 1
        if rc != 0:
            print("FAIL rc", rc, what); bad += 1; continue
        Yall = to_host(ybuf)
        Y = Yall[offy:offy + M * ldy].reshape(M, ldy)
        Xc = np.asfortranarray(Xh[:, :rows].T)
        model = oracle.MODEL_ASCENDING_FMA if op.ascending_k else oracle.MODEL_KERNEL
        want = oracle.model_apply(A, Xc, model)
        ok = np.array_equal(np.asfortranarray(Y[:, :rows].T), want)
        pad_ok = bool(np.all(Y[:, rows:] == -7.25)) and bool(np.all(Yall[:offy] == -7.25)) and bool(np.all(Yall[offy + M * ldy:] == -7.25))
        tip_ok = (not use_tips) or np.array_equal(to_host(tips), want[0])
        host_ok = True
        if rng.random() < 0.25:
            crumb(f"case {case}: apply_host X@{Xc.ctypes.data:#x} ({Xc.nbytes} bytes)")
            host = op.apply_host(Xc)
            host_ok = np.array_equal(host, want)
        integ_ok = True
        if which == sri.OP_DNN_INV and rng.random() < 0.7:
            # sri_integrate: Y = Dinv B - w x_base^T with w = Dinv Din.  When the last k chunk has a free slot (rows % 16 != 0)
            # the term rides in the product: the FMA chain over [Dinv | -w] and [B; x_base].  Otherwise it is the kernel's
            # epilogue: the chain over Dinv and B, then ONE fused multiply-add — the same chain, one link longer.
            xb = rng.uniform(-1, 1, M)
            dxb = to_dev(xb)
            ybuf.fill_(-7.25)
            crumb(f"case {case}: integrate")
            rc = lib.sri_integrate(op._h, xp, ldx, dxb.data_ptr(), yp, ldy, M, s)
            torch.cuda.synchronize()
            Yi = to_host(ybuf)[offy:offy + M * ldy].reshape(M, ldy)
            w = np.zeros(rows)
            for k in range(rows):
                w = w + Dinv[:, k] * Din[k]
            # (either form — folded into the product, or the store stage's fused multiply-add — is the chain over the extended operands)
            Aext = np.asfortranarray(np.concatenate([Dinv, -w[:, None]], axis=1))
            Xext = np.asfortranarray(np.concatenate([Xc, xb[None, :]], axis=0))
            want_i = oracle.model_apply(Aext, Xext, model)
            integ_ok = rc == 0 and np.array_equal(np.asfortranarray(Yi[:, :rows].T), want_i) and bool(np.all(Yi[:, rows:] == -7.25))
        if not (ok and pad_ok and tip_ok and host_ok and integ_ok):
            bad += 1
            print("FAIL", what, "bits", ok, "padding", pad_ok, "tips", tip_ok, "host", host_ok, "integrate", integ_ok)
        op.close()
    print("kernels exercised:", kernels)
    print(f"{a.cases} cases, {bad} failures" + ("" if bad else "  — FUZZ CLEAN"))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
