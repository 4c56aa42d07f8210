#!/usr/bin/env python3
"""Round 5, the reference's literal layout (255 rows, ld = 255) on a page-aligned BASELINE-sized panel, which ENDS on a page
boundary and therefore runs the edge-safe odd-row-count kernel (0.837 of the peak): is it worth launching the panel's last
column tile on its own, so that everything in front of it takes the LEAN kernel?  Measured with two public calls, no library
change: A = one sri_apply over all M columns; B = sri_apply over columns [0, M - tail) (an interior range: the bytes behind
its last column are the panel's own, and it ends inside a page -> lean) + sri_apply over the last `tail` columns (edge-safe;
the column plan gives such a small panel to 64 x 64 tiles).  Same bits.  One process, interleaved rounds."""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--rods", type=int, default=65536)
    ap.add_argument("--tails", default="128,256,1024")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    import torch
    import experimental_gpu_programming_for_a_spectral_numerical_integration_amd as sri

    dev = torch.device("cuda", 0)
    op = sri.SpectralApply(a.n, sri.OP_DNN_INV)
    rows, M = op.rows, 4 * a.rods
    lib = sri.load_library()
    X = torch.rand((M, rows), dtype=torch.float64, device=dev) * 2 - 1          # ld = rows: the literal layout
    assert X.data_ptr() % 4096 == 0
    ends_on_page = (X.data_ptr() + M * rows * 8) % 4096 == 0
    Y0, Y1 = torch.empty_like(X), torch.empty_like(X)
    s = torch.cuda.current_stream().cuda_stream

    def whole():
        assert lib.sri_apply(op._h, X.data_ptr(), rows, Y0.data_ptr(), rows, M, s) == 0

    def split(tail):
        def f():
            head = M - tail
            assert lib.sri_apply(op._h, X.data_ptr(), rows, Y1.data_ptr(), rows, head, s) == 0
            assert lib.sri_apply(op._h, X.data_ptr() + head * rows * 8, rows, Y1.data_ptr() + head * rows * 8, rows, tail, s) == 0
        return f

    variants = {"one launch": whole}
    for t in [int(v) for v in a.tails.split(",")]:
        variants[f"head + last {t} columns"] = split(t)
    names = {}
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        for _ in range(20):
            whole()
        torch.cuda.synchronize()
    times = {k: [] for k in variants}
    for _ in range(a.rounds):
        for k, fn in variants.items():
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / a.reps * 1e3)
            names[k] = op.last_kernel
    flops = 2.0 * rows * rows * M
    print(f"N={a.n} rows={rows} ld={rows} rods={a.rods}: panel ends on a page boundary: {ends_on_page}")
    for k in variants:
        t = min(times[k])
        print(f"  {k:32s} {t:8.1f} us (median {statistics.median(times[k]):8.1f})  {flops / t / 1e6 / 78.6:.4f} of the peak   last kernel {names[k]}")
    print("  same bits:", torch.equal(Y0, Y1))


if __name__ == "__main__":
    main()
