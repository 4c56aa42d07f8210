#!/usr/bin/env python3
"""A/B timing of two builds of libsri_hip.so in ONE process, interleaved rounds (cdna_hip_programming.md §5.4 rule 24).

    python tools/ab_apply.py --n 256 --rods 65536 --libs A=path/to/a.so B=path/to/b.so [--rounds 5] [--reps 200]
    python tools/ab_apply.py --n 64 --rods 4096 --libs small=lib.so general=lib.so,opt:4=0        (sri_option 4 = SRI_OPT_SMALL_KERNEL)
    python tools/ab_apply.py --n 256 --rods 1024 --libs own=lib.so mid=lib.so,SRI_DEBUG=force_geo=m  (environment at handle creation)

Prints min / median kernel time per variant (HIP events around `reps` back-to-back launches) and the TF/s."""
import argparse
import ctypes as C
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
    ap.add_argument("--operator", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--data", default="rod")
    ap.add_argument("--ld", type=int, default=0, help="leading dimension of X and Y (default: rows)")
    ap.add_argument("--libs", nargs="+", required=True)
    a = ap.parse_args()
    import torch
    import bench

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    vp, u32, sz, i32 = C.c_void_p, C.c_uint32, C.c_size_t, C.c_int
    variants = []
    for spec in a.libs:
        name, path = spec.split("=", 1)
        envs = []
        if "," in path:                                        # name=path,ENV=VAL,...: environment seen by sri_apply_create
            path, *envs = path.split(",")
        opts = [kv[4:].split("=", 1) for kv in envs if kv.startswith("opt:")]    # opt:<sri_option number>=<value>: set after creation
        envs = [kv for kv in envs if not kv.startswith("opt:")]
        for kv in envs:
            k, v = kv.split("=", 1)
            os.environ[k] = v
        lib = C.CDLL(os.path.abspath(path))
        lib.sri_apply_create.argtypes = [C.POINTER(vp), u32, i32, i32]
        lib.sri_apply.argtypes = [vp, vp, sz, vp, sz, sz, vp]
        lib.sri_apply_dim.argtypes = [vp, C.POINTER(u32)]
        h = vp()
        rc = lib.sri_apply_create(C.byref(h), a.n, a.operator, 0)
        assert rc == 0, rc
        for k, v in opts:
            lib.sri_apply_set_option.argtypes = [vp, i32, C.c_long]
            assert lib.sri_apply_set_option(h, int(k), int(v)) == 0, (k, v)
        for kv in envs:
            os.environ.pop(kv.split("=", 1)[0], None)
        r = u32()
        lib.sri_apply_dim(h, C.byref(r))
        variants.append((name, lib, h, r.value))
    rows = variants[0][3]
    X = bench.device_panel(torch, a.n, rows, a.rods, a.data, 0x5EED, dev)
    M = 4 * a.rods
    ld = a.ld or rows
    if ld != rows:                                             # strided panels: ld doubles per column, the tail is NaN
        Xs = torch.full((M, ld), float("nan"), dtype=torch.float64, device=dev)
        Xs[:, :rows] = X
        X = Xs
    Ys = [torch.empty_like(X) for _ in variants]
    stream = torch.cuda.current_stream().cuda_stream

    def run(k, reps):
        name, lib, h, _ = variants[k]
        for _ in range(reps):
            rc = lib.sri_apply(h, X.data_ptr(), ld, Ys[k].data_ptr(), ld, M, stream)
            assert rc == 0

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:                     # clock ramp
        run(0, 20)
        torch.cuda.synchronize()
    times = {v[0]: [] for v in variants}
    for _ in range(a.rounds):
        for k, v in enumerate(variants):
            run(k, 10)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(k, a.reps)
            e1.record()
            torch.cuda.synchronize()
            times[v[0]].append(e0.elapsed_time(e1) / a.reps)
    flops = 8.0 * rows * rows * a.rods
    same = all(torch.equal(Ys[0][:, :rows], y[:, :rows]) for y in Ys[1:])
    for name, ts in times.items():
        print(f"{name:>10s}  N={a.n} op={a.operator} ld={ld} rods={a.rods}  min {min(ts):.4f} ms  median {statistics.median(ts):.4f} ms  "
              f"{flops / min(ts) / 1e9:.2f} TF/s (min)  {flops / statistics.median(ts) / 1e9:.2f} TF/s (median)")
    print("outputs bit-identical across variants:", same)


if __name__ == "__main__":
    main()
