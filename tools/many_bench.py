#!/usr/bin/env python3
"""sri_apply_many against a loop of sri_apply: `count` panels of BASELINE config 2's size (N = 64, 4 096 rods) per launch.

    python tools/many_bench.py [--n 64] [--rods 4096] [--count 32] [--reps 200]

Prints one JSON line: algorithmic bytes / time for both forms, as a fraction of 8 TB/s (and of the 5.6 TB/s a pure copy of
the same byte mix reaches on this device)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--rods", type=int, default=4096)
    ap.add_argument("--count", type=int, default=32)
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    import torch
    import bench
    import experimental_gpu_programming_for_a_spectral_numerical_integration_amd as sri

    dev = torch.device("cuda", 0)
    op = sri.SpectralApply(a.n, sri.OP_DN)
    rows, M = op.rows, 4 * a.rods
    Xs = [bench.device_panel(torch, a.n, rows, a.rods, "rod", 0x5EED + i, dev) for i in range(a.count)]
    Ys = [torch.empty_like(X) for X in Xs]
    Y1 = [torch.empty_like(X) for X in Xs]
    lib, s = sri.load_library(), torch.cuda.current_stream().cuda_stream
    px = torch.tensor([X.data_ptr() for X in Xs], dtype=torch.int64).to(dev)
    py = torch.tensor([Y.data_ptr() for Y in Ys], dtype=torch.int64).to(dev)

    def many():
        rc = lib.sri_apply_many(op._h, px.data_ptr(), rows, py.data_ptr(), rows, M, a.count, s)
        assert rc == 0, rc

    def singles():
        for X, Y in zip(Xs, Y1):
            rc = lib.sri_apply(op._h, X.data_ptr(), rows, Y.data_ptr(), rows, M, s)
            assert rc == 0, rc

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:                   # clock ramp
        many()
        torch.cuda.synchronize()
    out = {}
    for name, fn in (("many", many), ("singles", singles)):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / a.reps)
        out[name] = best
    same = all(torch.equal(y, z) for y, z in zip(Ys, Y1))
    alg = a.count * (64.0 * rows * a.rods) + 8.0 * rows * rows
    line = {"tool": "many_bench", "N": a.n, "rods_per_panel": a.rods, "panels": a.count, "kernel": op.last_kernel,
            "algorithmic_bytes": alg, "bit_identical_to_single_applies": same,
            "many_ms": out["many"], "many_GBps": alg / out["many"] / 1e6, "many_frac_of_8TBps": alg / out["many"] / 1e6 / 8000,
            "many_frac_of_measured_copy_5600GBps": alg / out["many"] / 1e6 / 5600,
            "singles_ms": out["singles"], "singles_GBps": alg / out["singles"] / 1e6, "singles_frac_of_8TBps": alg / out["singles"] / 1e6 / 8000,
            "rod_states_per_s_many": a.count * a.rods / out["many"] * 1e3, "rod_states_per_s_singles": a.count * a.rods / out["singles"] * 1e3}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
