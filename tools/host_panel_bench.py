"""PCIe-inclusive rate of the host-panel entry points (sri_apply_host: what the drop-in header calls).

    python tools/host_panel_bench.py [--n 256] [--rods 65536] [--reps 5]

Prints one JSON line per mode: wall time per call (first call = cold: building the handle's pinned staging ring and device
slots), the bytes each way, and the link rate that corresponds to — to be read against tools/host_transfer_probe.hip (what
the link gives with nothing to compute) and the 0.5 ms the apply itself takes.  `--threads a,b,c` sweeps the number of CPU
copy threads of the staged path (plain numpy memory, the default of the library); `--pinned` adds the opt-in path for
callers whose panels are page-locked (torch's pinned allocator stands in for hipHostMalloc).  Never what bench.py reports.
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
    ap.add_argument("--threads", type=str, default="-1")
    ap.add_argument("--pinned", action="store_true")
    ap.add_argument("--block-cols", type=str, default="", help="comma list: override the library's block size (diagnostics)")
    args = ap.parse_args()
    import experimental_gpu_programming_for_a_spectral_numerical_integration_amd as sri

    M = 4 * args.rods
    rng = np.random.default_rng(0x5EED)

    def measure(op, X, Y, label, extra):
        times = []
        for _ in range(args.reps + 1):
            t = time.perf_counter()
            op.apply_host(X, out=Y)
            times.append(time.perf_counter() - t)
        gb = X.nbytes / 1e9
        warm = sorted(times[1:])
        print(json.dumps({"mode": label, "n": args.n, "rods": args.rods, "rows": op.rows, "panel_GB_each_way": gb,
                          "first_call_ms": times[0] * 1e3, "warm_ms_min": warm[0] * 1e3, "warm_ms_median": warm[len(warm) // 2] * 1e3,
                          "link_GBps_each_way": gb / warm[0], "rod_states_per_s": args.rods / warm[0], "kernel": op.last_kernel,
                          "host_cpus": os.cpu_count(), **extra}), flush=True)

    rows = sri.SpectralApply(args.n, args.which).rows
    X = np.asfortranarray(rng.uniform(-1, 1, size=(M, rows)).T)
    Y = np.empty((rows, M), order="F")
    for threads in [int(t) for t in args.threads.split(",")]:
        for bc in (args.block_cols.split(",") if args.block_cols else [""]):
            op = sri.SpectralApply(args.n, args.which)
            op.set_host_copy_threads(threads)
            if bc:
                op.set_option(sri.OPT_HOST_BLOCK_COLUMNS, int(bc))
            measure(op, X, Y, "staged (pageable caller memory)", {"copy_threads": threads, "block_cols": bc or "library"})
    if args.pinned:
        import torch

        Xp = torch.from_numpy(np.ascontiguousarray(X.T)).pin_memory()          # (M, rows) contiguous = column-major rows x M
        Yp = torch.empty_like(Xp).pin_memory()
        op = sri.SpectralApply(args.n, args.which)
        op.set_host_memory_pinned(True)
        measure(op, Xp.numpy().T, Yp.numpy().T, "caller-pinned (no staging)", {})
        assert np.array_equal(Yp.numpy().T, Y)


if __name__ == "__main__":
    main()
