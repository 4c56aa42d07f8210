#!/usr/bin/env python3
"""step_bench.py — sustained timing of the integration apply variants on one MI355X (after a clock ramp):
plain apply, apply + rank-1 initial-value term, and the fused Picard step (quaternion right-hand side formed in
the store stage), for panels with leading dimension m (columns 8-byte aligned when m is odd) and with the
leading dimension rounded up to a whole 128-byte line."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import experimental_gpu_programming_for_a_spectral_numerical_integration_amd as sri  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--rods", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=400)
    ap.add_argument("--ramp", type=float, default=1.0, help="seconds of clock ramp per variant")
    args = ap.parse_args()
    n, rods, m = args.n, args.rods, args.n - 1
    lib = sri.load_library()
    rod = sri.RodIntegrator(n)
    integ = sri.SpectralApply(n, sri.OP_DNN_INV)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    qe = (torch.rand((rods, 9), dtype=torch.float64, device=dev, generator=g) * 4 - 2)
    xb = torch.rand((rods * 4,), dtype=torch.float64, device=dev, generator=g)
    stream = torch.cuda.current_stream().cuda_stream
    out = {"n": n, "rods": rods, "reps": args.reps}
    for ld in (m, (m + 15) // 16 * 16):
        X = torch.rand((rods * 4, ld), dtype=torch.float64, device=dev, generator=g) * 2 - 1
        Y = torch.empty_like(X)
        cols = rods * 4
        calls = {
            "apply": lambda: lib.sri_apply(integ._h, X.data_ptr(), ld, Y.data_ptr(), ld, cols, stream),
            "integrate": lambda: lib.sri_integrate(integ._h, X.data_ptr(), ld, xb.data_ptr(), Y.data_ptr(), ld, cols, stream),
            "fused_step": lambda: lib.sri_rod_quaternion_step(rod._h, qe.data_ptr(), X.data_ptr(), ld, xb.data_ptr(), rods,
                                                              Y.data_ptr(), ld, None, None, None, stream),
        }
        for name, fn in calls.items():
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < args.ramp:     # clock ramp on this very kernel
                for _ in range(20):
                    assert fn() == 0
                torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / args.reps
            out[f"{name}_ld{ld}"] = {"ms": ms, "TFLOPs": 8.0 * m * m * rods / (ms * 1e-3) / 1e12,
                                     "GBps": 64.0 * m * rods / (ms * 1e-3) / 1e9}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
