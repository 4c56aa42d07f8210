#!/usr/bin/env python3
"""Back-to-back small applies: stream launches vs one HIP-graph replay of K applies (sri_apply is capturable: it only
enqueues a kernel).  Prints microseconds per apply for both.   python tools/graph_bench.py --n 64 --rods 4096 --k 100"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import experimental_gpu_programming_for_a_spectral_numerical_integration_amd as sri  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=64)
ap.add_argument("--rods", type=int, default=4096)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--panels", type=int, default=1, help="distinct (X, Y) pairs the applies cycle through")
a = ap.parse_args()
op = sri.SpectralApply(a.n, sri.OP_DN, device=0)
Xs = [torch.rand((4 * a.rods, a.n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(a.panels)]
Ys = [torch.empty_like(Xs[0]) for _ in range(a.panels)]


def burst():
    for i in range(a.k):
        op.apply(Xs[i % a.panels], out=Ys[i % a.panels])


t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    burst()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(20):
    e0.record(); burst(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / a.k * 1e3)
print(f"stream launches : {best:7.2f} us per apply (N={a.n}, {a.rods} rods, {op.last_kernel})")
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        burst()
ref = [y.clone() for y in Ys]
for y in Ys:
    y.zero_()
g.replay(); torch.cuda.synchronize()
assert all(torch.equal(y, r) for y, r in zip(Ys, ref))
best = 1e9
for _ in range(20):
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / a.k * 1e3)
print(f"graph replay    : {best:7.2f} us per apply ({a.k} applies per graph)")
