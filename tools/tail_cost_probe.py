#!/usr/bin/env python3
"""What do the odd-row-count (TAIL) variants cost, and what is the operator's DATA worth?  (GPU box.)

Times, interleaved in one process, on the same panel with leading dimension N:
  dn        D_N (N rows, plain kernel)
  inv       Dn_NN^-1 (N-1 rows: TAIL kernel)
  inv_pad   the same matrix padded with a zero row and column to N rows (plain kernel, same MFMA count as `inv`)
  dn_cut    D_N[:N-1, :N-1] = Dn_NN (N-1 rows: TAIL kernel, the differentiation operator's values)
`inv` against `inv_pad` is the cost of the tail machinery; `dn` against `inv_pad` the effect of the operator's values alone
(fp64 MFMA power, hence clock, depends on the operands).  Prints ms and TF/s (flops counted with the true row count)."""
import argparse
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--rods", type=int, default=65536)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--data", default="rod")
    a = ap.parse_args()
    import torch

    import bench
    import experimental_gpu_programming_for_a_spectral_numerical_integration_amd as sri

    n = a.n
    dev = torch.device("cuda", 0)
    lib = sri.load_library()
    inv = sri.SpectralApply(n, sri.OP_DNN_INV)
    D = np.asfortranarray(np.zeros((n, n)))
    assert lib.sri_cheb_dn(n, D.ctypes.data_as(sri._dp), n) == 0
    Ainv = np.asfortranarray(np.zeros((n - 1, n - 1)))
    Din = np.zeros(n - 1)
    assert lib.sri_cheb_integration(n, Ainv.ctypes.data_as(sri._dp), n - 1, Din.ctypes.data_as(sri._dp)) == 0
    pad = np.asfortranarray(np.zeros((n, n)))
    pad[: n - 1, : n - 1] = Ainv
    ops = {"dn": sri.SpectralApply(n, sri.OP_DN), "inv": inv, "inv_pad": sri.SpectralApply(n + 1, sri.OP_DN, matrix=pad),
           "dn_cut": sri.SpectralApply(n, sri.OP_DNN)}
    M = 4 * a.rods
    X = bench.device_panel(torch, n, n, a.rods, a.data, 0x5EED, dev)           # (M, n): ld = n for every operator
    Y = torch.empty_like(X)
    stream = torch.cuda.current_stream().cuda_stream

    def run(op, reps):
        for _ in range(reps):
            rc = lib.sri_apply(op._h, X.data_ptr(), n, Y.data_ptr(), n, M, stream)
            assert rc == 0

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        run(ops["dn"], 20)
        torch.cuda.synchronize()
    times = {k: [] for k in ops}
    for _ in range(a.rounds):
        for k, op in ops.items():
            run(op, 10)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(op, a.reps)
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / a.reps)
    for k, op in ops.items():
        rows = op.rows if k != "inv_pad" else n - 1
        fl = 8.0 * rows * rows * a.rods
        print(f"{k:>8s} N={n} rows={op.rows} kernel={op.last_kernel}  min {min(times[k]):.4f} ms  median {statistics.median(times[k]):.4f} ms  "
              f"{fl / min(times[k]) / 1e9:.2f} TF/s", flush=True)


if __name__ == "__main__":
    main()
