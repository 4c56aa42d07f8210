#!/usr/bin/env python3
"""One-off soak of the hand-synchronised kernels: many launches of each variant must be bit-identical to the first
(a race on the LDS ring or a mis-counted vmcnt shows up as run-to-run differences).  python tools/soak.py [--reps 300]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import experimental_gpu_programming_for_a_spectral_numerical_integration_amd as sri  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=300)
a = ap.parse_args()
g = torch.Generator(device="cuda").manual_seed(11)
bad = 0
noise = torch.rand((1 << 24,), dtype=torch.float64, device="cuda")
for n, which, M in [(256, 0, 33000), (256, 1, 20001), (1024, 0, 3000), (1024, 1, 1500), (64, 0, 16384), (64, 0, 3001), (64, 0, 70000),
                    (128, 0, 9000), (100, 0, 5000), (16, 0, 9999), (512, 0, 4000), (256, 2, 7777),
                    (320, 0, 9000), (384, 1, 5001), (640, 0, 3000), (513, 0, 2500), (897, 1, 1500),   # several 128-row blocks, both k orders
                    (257, 0, 9000), (258, 1, 7001), (1025, 0, 2100), (300, 0, 6000), (160, 0, 9000), (273, 1, 4000)]:   # round 4: strip, parts, lean twins
    op = sri.SpectralApply(n, which)
    X = torch.rand((M, op.rows), dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    tips = torch.empty(M, dtype=torch.float64, device="cuda")
    ref = op.apply(X, tips=tips).clone()
    rt = tips.clone()
    diffs = 0
    for i in range(a.reps):
        if i % 7 == 0:
            noise.mul_(1.0000001)
        Y = op.apply(X, tips=tips)
        if not (torch.equal(Y, ref) and torch.equal(tips, rt)):
            diffs += 1
    print(f"N={n} op={which} M={M} {op.last_kernel}: {diffs} of {a.reps} launches differ", flush=True)
    bad += diffs
for n, rods in [(256, 8192), (64, 20000), (16, 50000), (1024, 512), (48, 3000)]:
    rod = sri.RodIntegrator(n)
    qe = torch.rand((rods, 9), dtype=torch.float64, device="cuda", generator=g) * 4 - 2
    Q0, R0 = rod.integrate(qe)
    diffs = 0
    for i in range(max(a.reps // 10, 5)):
        Q, R = rod.integrate(qe)
        if not (torch.equal(Q, Q0) and torch.equal(R, R0)):
            diffs += 1
    print(f"rod pipeline N={n} rods={rods}: {diffs} runs differ ({rod.iterations} iterations)", flush=True)
    bad += diffs
torch.cuda.synchronize()
print("SOAK", "CLEAN" if bad == 0 else f"FAILED ({bad})")
sys.exit(0 if bad == 0 else 1)
