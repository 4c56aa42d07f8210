"""PCIe-inclusive rate of the host-panel entry points (sri_apply_host: what the drop-in header calls).

    python tools/host_panel_bench.py [--n 256] [--rods 65536] [--reps 5]

Prints one JSON line: wall time per call (first call = cold: page-locking the caller's buffers, building the slot ring),
the bytes each way, and the link rate that corresponds to — to be read against tools/host_transfer_probe.hip (what the
link gives with nothing to compute) and the 0.5 ms the apply itself takes.  Never what bench.py reports.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--rods", type=int, default=65536)
    ap.add_argument("--which", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import experimental_gpu_programming_for_a_spectral_numerical_integration_amd as sri

    op = sri.SpectralApply(args.n, args.which)
    M = 4 * args.rods
    rng = np.random.default_rng(0x5EED)
    X = np.asfortranarray(rng.uniform(-1, 1, size=(M, op.rows)).T)
    Y = np.empty((op.rows, M), order="F")
    times = []
    for _ in range(args.reps + 1):
        t = time.perf_counter()
        op.apply_host(X, out=Y)
        times.append(time.perf_counter() - t)
    gb = X.nbytes / 1e9
    warm = sorted(times[1:])
    print(json.dumps({"n": args.n, "rods": args.rods, "rows": op.rows, "panel_GB_each_way": gb,
                      "first_call_ms": times[0] * 1e3, "warm_ms_min": warm[0] * 1e3, "warm_ms_median": warm[len(warm) // 2] * 1e3,
                      "link_GBps_each_way": gb / warm[0], "rod_states_per_s": args.rods / warm[0], "kernel": op.last_kernel}))


if __name__ == "__main__":
    main()
