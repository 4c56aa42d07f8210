#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: rod-states/s of the fp64 D_N.X apply.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}] [--data {rod,uniform}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W  (N > 1, one rank per GPU, RCCL)

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the second command itself as a
CHILD process (before anything here touches the GPU; never an exec), relays rank 0's JSON line and exits
with the child's return code.

A "step" is one pass of the hot path over one batch of synthetic state panels already resident in
HBM: Y = D_N . X with N = 256 nodes and 65 536 rods (x 4 state columns) PER GPU — BASELINE.json
configs[2] at N=1, configs[3] at 8 GPUs (weak scaling on the batch axis).  `--config` selects the
other BASELINE configurations (2: N=64 / 4 096 rods, 5: N=1024 / 8 192 rods).  The panels are the
synthetic rod-strain fields north_star names (`--data rod`, SURVEY §8d (ii): per rod qe ~ U[-2,2)^9,
K = Phi(X) qe at the Chebyshev nodes, 4th column cos|K|) or U[-1,1) noise (`--data uniform`).
The apply needs no collective (rods are independent, the operator is replicated).  For N > 1 the job
ends with ONE RCCL all-gather over xGMI of the integrated tip state of every rod (north_star:
"all-gather ... only to reassemble the final integrated state"), inside the timed region; every apply
emits its tip state from the kernel epilogue at no extra traffic.  `--gather tip` gathers after every
step instead (on a side stream); the cost of one tip gather and of one full-panel gather are reported
separately.

Exit code: 0 = line printed and every check passed; 2 = the sampled columns fail the parity gate (the same gate the
GPU tests apply: oracle.parity_gate); 3 = the product's RCCL communicator could not be created or hung (the line is
printed from torch's RCCL and marked "degraded"); 4 = the gathered final state is wrong.  The line is printed in
every case and carries "exit_code".

Rank 0 prints ONE JSON line.  `roofline` is computed from ALGORITHMIC flops (8 N^2 per rod-state,
SURVEY.md §8d) and the kernel's mean launch duration measured here with HIP events on the launch
stream; `cpu_baseline` times the CPU oracle (test infrastructure, the checker only) on a bounded
sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = "experimental_gpu_programming_for_a_spectral_numerical_integration_amd"
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X datasheet dense fp64 matrix peak
MEASURED_FP64_MFMA_TFLOPS = 77.5   # what a bare MFMA stream sustains on this device (tools/mfma_f64_probe.hip, profiles/r01_mfma_f64_probe.log)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MEASURED_COPY_GBS = 5600.0     # a pure copy of the apply's byte mix on this device: 5.5-5.7 TB/s (tools/stream_probe.hip, profiles/r02_stream_probe.log)

# BASELINE.json configs (index = position in its "configs" list + 1): (N, rods per GPU)
CONFIGS = {2: (64, 4096), 3: (256, 65536), 4: (256, 65536), 5: (1024, 8192)}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--ramp-ms", type=float, default=1500.0,
                    help="untimed clock ramp before the W warmup steps: an idle MI355X sits at ~160 MHz sclk and needs "
                         "several hundred ms of load to reach its sustained 2.37 GHz (profiles/r01_power_clock.log)")
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=None,
                    help="BASELINE.json configuration: 2 = N=64/4096 rods, 3 = N=256/65536 rods (default), "
                         "4 = config 3 per GPU (pass --gpus 8), 5 = N=1024/8192 rods; --n/--rods override")
    ap.add_argument("--n", type=int, default=None, help="Chebyshev nodes N (default 256)")
    ap.add_argument("--rods", type=int, default=None, help="rods per GPU, 4 state columns each (default 65536)")
    ap.add_argument("--data", choices=["rod", "uniform"], default="rod",
                    help="rod = synthetic rod-strain fields (what north_star names), uniform = U[-1,1)")
    ap.add_argument("--operator", choices=["dn", "dnn_inv"], default="dn")
    ap.add_argument("--rotate", type=int, default=1,
                    help="number of distinct (X, Y) panel pairs the steps cycle through; config 2's 16.8 MB working "
                         "set is Infinity-Cache resident when timed back to back (SURVEY H5): --rotate 32 streams "
                         "it from HBM instead")
    ap.add_argument("--gather", choices=["final", "tip", "full", "none"], default="final",
                    help="exchange when N > 1: final = ONE all-gather of the integrated tip state after the K steps, "
                         "inside the timed region (default); tip = one per step, on a side stream; "
                         "full = whole output panel per step; none")
    ap.add_argument("--collective", choices=["capi", "torch"], default="capi",
                    help="who issues the RCCL all-gather: capi = libsri_hip.so's sri_allgather_* (default), "
                         "torch = torch.distributed")
    ap.add_argument("--reserve-cus", type=int, default=0,
                    help="CUs left free for an overlapped RCCL kernel (0 = none: with 8 tiles per workgroup a smaller "
                         "grid costs a whole extra round of tiles, measured +12%%)")
    ap.add_argument("--event-stride", type=int, default=8,
                    help="bracket every n-th launch of the timed region with a HIP event pair (kernel duration for the roofline)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (0 disables)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--ld", type=int, default=0,
                    help="leading dimension of the panels in doubles (default: the operator's row count; e.g. 256 for the 255-row "
                         "integration operator: columns on whole 128-byte lines, INTEGRATION.md §2).  Single-GPU runs only.")
    ap.add_argument("--long-run", type=int, default=500,
                    help="after the timed region (N = 1): at least this many back-to-back launches, reported as kernel_ms_long_run (0 disables)")
    args = ap.parse_args(argv)
    n0, r0 = CONFIGS[args.config if args.config is not None else 3]
    args.n = n0 if args.n is None else args.n
    args.rods = r0 if args.rods is None else args.rods
    return args


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n_gpus: int, argv, script: str = None) -> int:
    """`python bench.py --gpus N` (N > 1) outside torchrun: start one rank per GPU as a CHILD
    torch.distributed.run job, relay its stdout (rank 0's single JSON line) and return its rc.
    Runs before this process has imported torch or touched the GPU; it never execs."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=os.getcwd())
    out, _ = proc.communicate()
    # rank 0's result line: the last line that holds a JSON object (two ranks share this pipe: should another rank's chatter
    # ever land on the same line, the object is still taken from its opening brace)
    lines, rest = [], []
    for ln in out.decode(errors="replace").splitlines():
        at = ln.find('{"')
        obj = None
        if at >= 0:
            try:
                obj, _ = json.JSONDecoder().raw_decode(ln[at:])
            except ValueError:
                obj = None
        if isinstance(obj, dict):
            lines.append(json.dumps(obj))
            if ln[:at].strip():
                rest.append(ln[:at])
        elif ln.strip():
            rest.append(ln)
    for ln in rest:                                        # launcher chatter is not part of the contract
        print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    if proc.returncode == 0 and not lines:
        print("bench.py: ranks exited 0 without a result line", file=sys.stderr)
        return 1
    return proc.returncode


def lib_sources_sha16() -> str:
    """sha256 (first 16 hex digits) over the sources libsri_hip.so is built from — what a committed PMC summary
    (profiles/hbm_traffic_*.json, written by tools/summarize_profile.py) is keyed on: a summary taken with other kernel
    sources is replayed as `traffic_stale: true` and never prices anything."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, PKG, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".h")) or f == "Makefile")
    files.append(os.path.join(ROOT, "include", "sri_capi.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _stats(ts):
    ts = sorted(ts)
    return ts[0], ts[len(ts) // 2]


def usable_cores() -> int:
    """Host threads this process may actually run on: the smaller of the affinity mask and the cgroup CPU quota (a GPU box
    hands one job a share of its cores; 128 OpenMP threads on a 16-core share only measure the scheduler)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                       # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:                                                                                    # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(n: int, budget_s: float, data: str, max_rods: int = 65536):
    """Oracle (kind "port") on a bounded sample of the same workload: single thread like the
    reference's main.cpp, then all host cores; >= 3 repetitions each, min and median."""
    import oracle

    A = oracle.dn(n)
    probe_rods = 64
    Xp = host_panel(n, probe_rods, data, 0x5EED)
    oracle.apply(A, Xp)                                   # warm
    t0 = time.perf_counter(); oracle.apply(A, Xp); t = time.perf_counter() - t0
    rate = probe_rods / max(t, 1e-9)
    reps = 5
    rods = int(min(max(rate * budget_s * 0.6 / reps, probe_rods), max_rods))   # at most the workload itself
    X = host_panel(n, rods, data, 0x5EED)
    t1 = []
    for _ in range(reps):
        t0 = time.perf_counter(); oracle.apply(A, X); t1.append(time.perf_counter() - t0)
    threads = min(oracle.max_threads(), usable_cores())
    # all cores: a sample `threads/2` times larger so that a repetition is not shorter than the fork/join noise
    rods_mt = int(min(rods * max(threads // 2, 1), 65536, (512 << 20) // (32 * n)))
    Xm = host_panel(n, rods_mt, data, 0x5EED)
    oracle.apply(A, Xm, threads=threads)                  # warm the thread pool
    tm = []
    for _ in range(reps):
        t0 = time.perf_counter(); oracle.apply(A, Xm, threads=threads); tm.append(time.perf_counter() - t0)
    b1, m1 = _stats(t1)
    bm, mm = _stats(tm)
    return {
        "value": rods / b1, "unit": "rod-states/s", "cores": 1, "kind": "port",
        "sample": f"oracle/sri_oracle.c single thread, {rods} rods x 4 cols at N={n}, {data} panels, "
                  f"min of {reps} (median in median_value)",
        "median_value": rods / m1, "reps": reps,
        "all_cores_value": rods_mt / bm, "all_cores_median_value": rods_mt / mm, "all_cores": threads,
        "all_cores_sample": f"{rods_mt} rods, OpenMP over column blocks on {threads} threads (affinity mask / cgroup quota; "
                            f"the host reports {os.cpu_count()} logical CPUs), min of {reps}",
        "eigen": eigen_baseline(n, rods),
    }


def eigen_baseline(n: int, rods: int):
    """SURVEY §8(d) CPU plan item 3: if a REAL <Eigen/Dense> is installed on this box (never this repo's compat/
    stand-ins), time the reference's own library doing the reference's own product — Eigen's `A * X`, one thread — as the
    "reference-library" figure.  Returns {"available": False, ...} where there is none (this pool's image)."""
    import shutil
    import tempfile

    cands = [os.environ.get("EIGEN3_INCLUDE_DIR"), "/usr/include/eigen3", "/usr/local/include/eigen3", "/opt/homebrew/include/eigen3",
             "/usr/include", "/usr/local/include"]
    inc = next((c for c in cands if c and os.path.exists(os.path.join(c, "Eigen", "Dense"))
                and os.path.realpath(c) != os.path.realpath(os.path.join(ROOT, "compat"))), None)
    gxx = shutil.which("g++")
    if inc is None or gxx is None:
        return {"available": False, "searched": [c for c in cands if c]}
    try:
        # (NOT oracle/_ref/: that name is reserved for a build of the reference's own sources, which this image cannot produce)
        outdir = os.path.join(ROOT, "build", "eigen_baseline")
        os.makedirs(outdir, exist_ok=True)
        exe = os.path.join(outdir, "eigen_baseline")
        subprocess.check_call([gxx, "-O3", "-march=native", "-std=c++17", "-I" + inc, os.path.join(ROOT, "oracle", "eigen_baseline.cpp"), "-o", exe],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        r = subprocess.run([exe, str(n), str(min(rods, 2048)), "5"], capture_output=True, text=True, timeout=300)
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        return {"available": True, "value": rec["rod_states_per_s"], "unit": "rod-states/s", "cores": 1, "kind": "reference-library",
                "sample": f"Eigen {rec['eigen_version']} A*X, one thread, {rec['rods']} rods x 4 cols at N={n} (include dir {inc})"}
    except Exception as e:                                  # noqa: BLE001 — a baseline that cannot be built is reported, not fatal
        return {"available": False, "error": str(e)[:200]}


def host_panel(n: int, rods: int, data: str, seed: int) -> np.ndarray:
    """Column-major n x (4*rods) synthetic panel on the host (cpu_baseline sample)."""
    rng = np.random.default_rng(seed)
    if data == "uniform":
        return np.asfortranarray(rng.uniform(-1, 1, size=(rods * 4, n)).T)
    x = 0.5 * (1 + np.cos(np.pi * np.arange(n) / (n - 1)))
    t = 2 * x - 1
    P = np.stack([np.ones_like(t), t, 0.5 * (3 * t * t - 1)], axis=1)
    qe = rng.uniform(-2.0, 2.0, size=(rods, 3, 3))
    K = np.einsum("ne,rae->ran", P, qe)                                    # rod, comp, node
    fourth = np.cos(np.sqrt((K * K).sum(axis=1, keepdims=True)))
    return np.asfortranarray(np.concatenate([K, fourth], axis=1).reshape(rods * 4, n).T)


def device_panel(torch, n: int, rows: int, rods: int, data: str, seed: int, dev):
    """(4*rods, rows) torch panel = column-major rows x 4*rods, generated on the device.

    rod: per rod qe ~ U[-2,2)^9 (3 strain components x 3 Legendre modes, utilities.h:49-67), K(X_i) =
    Phi(X_i) qe at the first `rows` Chebyshev nodes (chebyshev_differentiation.h:26), 4th column cos|K|."""
    g = torch.Generator(device=dev).manual_seed(seed)
    if data == "uniform":
        return torch.rand((4 * rods, rows), dtype=torch.float64, device=dev, generator=g) * 2 - 1
    j = torch.arange(rows, dtype=torch.float64, device=dev)
    t = torch.cos(torch.pi * j / (n - 1))                                   # 2x - 1 at the nodes
    P = torch.stack([torch.ones_like(t), t, 0.5 * (3 * t * t - 1)], dim=0)  # 3 x rows
    X = torch.empty((rods, 4, rows), dtype=torch.float64, device=dev)
    step = 16384
    for r0 in range(0, rods, step):
        r1 = min(r0 + step, rods)
        qe = torch.rand((r1 - r0, 3, 3), dtype=torch.float64, device=dev, generator=g) * 4 - 2
        K = qe @ P                                                          # rod, comp, node
        X[r0:r1, :3] = K
        X[r0:r1, 3] = torch.cos(torch.linalg.vector_norm(K, dim=1))
    return X.reshape(4 * rods, rows)


# ---- exit codes (after the JSON line has been printed) -------------------------------------------------------------
EXIT_OK = 0
EXIT_PARITY = 2      # the sampled columns fail the parity gate (oracle.parity_gate == tests/conftest.assert_apply_parity)
EXIT_COMM = 3        # the product's RCCL communicator (sri_comm_*) could not be created / hung: the line is degraded
EXIT_GATHER = 4      # the gathered final state does not hold every rank's slab at its offset


def ghost_tips(rank: int, M: int, torch):
    """Rehearsal only: the tip state a GPU-less stand-in rank contributes — a pattern rank 0 can recompute."""
    return torch.arange(M, dtype=torch.float64) * 2.0 ** -30 + float(rank + 1)


def slab_report(tips_all, world: int, M: int, gpu_ranks: int, torch):
    """Rank 0, after the final gather: every rank's slab must sit at its own offset r*M — ghost slabs equal their pattern,
    GPU slabs are finite and pairwise different (every rank seeds its panel with its rank)."""
    bad = []
    flat = tips_all.detach().cpu()
    for r in range(world):
        slab = flat[r * M:(r + 1) * M]
        if r >= gpu_ranks:
            if not torch.equal(slab, ghost_tips(r, M, torch)):
                bad.append(r)
        elif not bool(torch.isfinite(slab).all()) or (r > 0 and torch.equal(slab, flat[0:M])):
            bad.append(r)
    return bad


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))     # before torch / the GPU are touched
    sys.exit(run_rank(args, world_env))


def run_rank(args, world_env) -> int:
    import torch

    # Libraries (RCCL's version banner, ...) write to stdout; the contract is ONE JSON line there.  Route
    # fd 1 to stderr until the result is printed.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    # SRI_BENCH_REHEARSAL=1: rehearse N > 1 on a ONE-GPU box — every rank uses cuda:0 and the collectives are
    # staged through CPU tensors over gloo (RCCL refuses two ranks on one device).  Timings are meaningless
    # there; the point is to run the rank / world-size / offset logic for real, at the real per-rank shape.
    # A GPU box admits only 6 processes on its card, so at most SRI_BENCH_REHEARSAL_GPU_RANKS ranks (default 4) run the
    # apply; the ranks beyond are GHOSTS: they never touch the GPU, take part in every collective in the same order
    # and contribute a recomputable pattern as their tip state — which is how `--gpus 8` is rehearsed on one device.
    rehearsal = os.environ.get("SRI_BENCH_REHEARSAL") == "1"
    gpu_ranks = min(world, max(1, int(os.environ.get("SRI_BENCH_REHEARSAL_GPU_RANKS", "4")))) if rehearsal else world
    ghost = rehearsal and rank >= gpu_ranks
    inject = os.environ.get("SRI_BENCH_INJECT", "")         # test hook: comm_fail | comm_hang (tests/test_bench_contract_gpu.py)
    if rehearsal:
        local_rank = 0
    dev = torch.device("cpu") if ghost else torch.device("cuda", local_rank)
    if not ghost:
        torch.cuda.set_device(local_rank)
    dist = None
    # SRI_BENCH_FORCE_DIST=1: run the N > 1 code path (RCCL init, side-stream tip gather, full-panel
    # gather) with whatever world size there is, even 1 — used to rehearse it on a one-GPU box.
    use_dist = world > 1 or os.environ.get("SRI_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    comm_box = [None]                                       # the C ABI's RCCL communicator (created below)
    comm_hung = False
    degraded = None                                         # why the data-path collective is not the product's own

    def all_gather(out, inp):
        if rehearsal:
            tmp = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(tmp, inp.cpu())
            out.copy_(tmp)
        elif comm_box[0] is not None:
            if inp.dim() == 1:
                comm_box[0].allgather_tips(inp, out=out)     # sri_allgather_tips on torch's current stream
            else:
                comm_box[0].allgather_panel(inp, out=out)    # sri_allgather_panel
        else:
            dist.all_gather_into_tensor(out, inp)

    def all_reduce(t, op):
        if rehearsal:
            c = t.cpu()
            dist.all_reduce(c, op=op)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=op)

    def sync():
        if not ghost:
            torch.cuda.synchronize()

    def barrier():
        if rehearsal:
            # (dist.barrier() initialises the GPU runtime even on a gloo group — measured: it opens /dev/kfd — which a
            # ghost rank must never do; an all-reduce of one CPU word is the same rendezvous)
            dist.all_reduce(torch.zeros(1, dtype=torch.float64))
        else:
            dist.barrier()

    def fence():
        sync()
        if use_dist:
            barrier()
            sync()

    import experimental_gpu_programming_for_a_spectral_numerical_integration_amd as sri

    if not os.path.exists(sri.library_path()):
        raise SystemExit("libsri_hip.so missing: run __graft_entry__.build() first (no CPU fallback)")

    # The data-path exchange goes through the product's C ABI (sri_comm_*: RCCL called from C++ on the launch
    # stream); torch.distributed keeps the contract's barrier / max-over-ranks plumbing and ships the 128-byte
    # rendezvous token.  Two phases, so that a rank that cannot even load RCCL never leaves the others waiting inside the
    # collective ncclCommInitRank: (1) every rank probes locally (sri_comm_unique_id: dlopen + symbol binding) and the
    # ranks agree; (2) only then is the communicator created, on a watchdog thread.  If that fails or hangs on ANY rank,
    # the measurement still runs — on torch's RCCL, so that a number exists — but the line says "degraded", every failing
    # rank and its error are named on stderr, and the process exits EXIT_COMM: a broken communicator never ends in rc 0.
    collective = "none"
    if use_dist and rehearsal:
        collective = f"gloo rehearsal (CPU staging; {gpu_ranks} GPU ranks + {world - gpu_ranks} ghost ranks on one device)"
    elif use_dist and args.collective == "torch":
        collective = "torch.distributed (RCCL)"
    elif use_dist:
        import threading
        from experimental_gpu_programming_for_a_spectral_numerical_integration_amd import parallel
        err = ""
        try:
            if inject == "comm_probe_fail" and rank == world - 1:
                raise RuntimeError("injected: RCCL could not be loaded")
            probe_id = parallel.Communicator.unique_id()
        except Exception as e:                              # noqa: BLE001 — named in the result line and on stderr
            err, probe_id = f"probe: {e}", None
        okf = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if okf.item() == 1.0:
            box = [probe_id if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            err_box = [""]

            def create():                                   # ncclCommInitRank is collective: never wait on it forever
                try:
                    torch.cuda.set_device(local_rank)
                    if inject == "comm_fail" and rank == world - 1:
                        raise RuntimeError("injected: sri_comm_create failed")
                    if inject == "comm_hang" and rank == world - 1:
                        time.sleep(1e6)
                    comm_box[0] = parallel.Communicator(box[0], rank, world, local_rank)
                except Exception as e:                      # noqa: BLE001
                    err_box[0] = f"create: {e}"

            th = threading.Thread(target=create, daemon=True)
            th.start()
            th.join(float(os.environ.get("SRI_COMM_TIMEOUT_S", "120")))
            if th.is_alive():
                err_box[0] = f"create: sri_comm_create did not return within {os.environ.get('SRI_COMM_TIMEOUT_S', '120')} s"
                comm_hung = True
            err = err_box[0]
            okf = torch.tensor([1.0 if (comm_box[0] is not None and not comm_hung) else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if okf.item() == 1.0:
            collective = "sri_comm (libsri_hip.so: RCCL from C++)"
        else:
            errs = [None] * world
            dist.all_gather_object(errs, err)
            failing = [f"rank {r}: {e}" for r, e in enumerate(errs) if e]
            if err:
                print(f"bench.py: rank {rank}: sri_comm unavailable — {err}", file=sys.stderr, flush=True)
            comm_box[0] = None
            degraded = "sri_comm unavailable — " + ("; ".join(failing) if failing else "no rank reported an error")
            collective = "torch.distributed (RCCL); DEGRADED: " + degraded

    n, rods = args.n, args.rods
    which = sri.OP_DN if args.operator == "dn" else sri.OP_DNN_INV
    if ghost:
        op, rows = None, (n if which == sri.OP_DN else n - 1)
    else:
        op = sri.SpectralApply(n, which, device=local_rank)
        rows = op.rows
    M = 4 * rods
    reserve = max(args.reserve_cus, 0)
    if reserve > 0 and not ghost:
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        op.set_max_workgroups(max(cus - reserve, 1))

    nrot = max(args.rotate, 1)
    if ghost:
        Xs = Ys = [None] * nrot
        X = Y = None
    else:
        Xs = [device_panel(torch, n, rows, rods, args.data, 0x5EED + rank + 1000 * r, dev) for r in range(nrot)]
        if args.ld > rows:                                  # strided panels: `ld` doubles per column, NaN between the columns
            def padded(t):
                buf = torch.full((t.shape[0], args.ld), float("nan"), dtype=torch.float64, device=dev)
                buf[:, :rows] = t
                return buf[:, :rows]
            Xs = [padded(x) for x in Xs]
        Ys = [torch.empty_like(Xs[0]) if args.ld <= rows else torch.empty((M, args.ld), dtype=torch.float64, device=dev)[:, :rows]
              for _ in range(nrot)]
        X, Y = Xs[0], Ys[0]
    ld = args.ld if args.ld > rows else rows
    if ld != rows and use_dist:
        raise SystemExit("bench.py: --ld is a single-GPU option")

    # exchange step (N > 1): tips of step i gathered on a side stream under the apply of step i+1
    comm_stream = torch.cuda.Stream(device=dev) if (use_dist and not ghost) else None
    tips_local = tips_all = None
    if use_dist:
        tips_local = [ghost_tips(rank, M, torch) if ghost else torch.empty(M, dtype=torch.float64, device=dev) for _ in range(2)]
        tips_all = [torch.empty(M * world, dtype=torch.float64, device=dev) for _ in range(2)]
    full_all = torch.empty((M * world, rows), dtype=torch.float64, device=dev) if (use_dist and args.gather == "full") else None
    ready = [torch.cuda.Event() for _ in range(2)] if not ghost else None
    gathered = [torch.cuda.Event() for _ in range(2)] if not ghost else None

    def timing_event():
        return None if ghost else torch.cuda.Event(enable_timing=True)

    def record(ev):
        if ev is not None:
            ev.record()

    def elapsed_ms(a, b):
        return 0.0 if a is None else a.elapsed_time(b)

    # (a ghost rank must not even LOAD the library: registering its code objects opens the GPU, and the box counts that)
    lib = None if ghost else sri.load_library()
    stream0 = None if ghost else torch.cuda.current_stream().cuda_stream

    def step(i: int, ev_pair=None):
        if ghost:
            if use_dist and args.gather == "tip":
                all_gather(tips_all[i & 1], tips_local[i & 1])
            elif use_dist and args.gather == "full":
                all_gather(full_all, torch.zeros((M, rows), dtype=torch.float64))
            return
        if ev_pair is None and (not use_dist or args.gather in ("final", "none")):
            # no per-launch events, no per-step exchange: the bare C-ABI call (a few-microsecond kernel is launch-bound
            # otherwise, and N > 1 must not pay more per step than N = 1 does)
            k = i % nrot
            if use_dist and args.gather == "final":
                rc = lib.sri_apply_tips(op._h, Xs[k].data_ptr(), ld, Ys[k].data_ptr(), ld, M, tips_local[i & 1].data_ptr(), stream0)
            else:
                rc = lib.sri_apply(op._h, Xs[k].data_ptr(), ld, Ys[k].data_ptr(), ld, M, stream0)
            if rc != 0:
                raise sri.SriError(rc, "sri_apply")
            return
        b = i & 1
        Xi, Yi = Xs[i % nrot], Ys[i % nrot]
        tip_mode = use_dist and args.gather in ("tip", "final")
        if use_dist and args.gather == "tip":
            torch.cuda.current_stream().wait_event(gathered[b])       # tip buffer b free again
        if ev_pair is not None:
            ev_pair[0].record()
        if ld != rows:                                          # (the Python mirror takes contiguous panels)
            rc = lib.sri_apply(op._h, Xi.data_ptr(), ld, Yi.data_ptr(), ld, M, stream0)
            if rc != 0:
                raise sri.SriError(rc, "sri_apply")
        else:
            op.apply(Xi, out=Yi, tips=tips_local[b] if tip_mode else None)  # the epilogue emits the tip state itself
        if ev_pair is not None:
            ev_pair[1].record()
        if use_dist and args.gather == "tip":
            ready[b].record()
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(ready[b])
                all_gather(tips_all[b], tips_local[b])
                gathered[b].record()
        elif use_dist and args.gather == "full":
            all_gather(full_all, Yi)

    # untimed clock ramp (the apply alone — no collective, so ranks need not agree on its length),
    # then the W warmup steps of the contract
    t_ramp = time.perf_counter()
    while not ghost and (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        for _ in range(20):
            if ld != rows:
                lib.sri_apply(op._h, X.data_ptr(), ld, Y.data_ptr(), ld, M, stream0)
            else:
                op.apply(X, out=Y)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    if use_dist and args.gather != "none":
        # the data-path collective once, untimed: RCCL sets up its channels on a communicator's first collective
        # (hundreds of ms), which would otherwise land in the timed region of `--gather final`
        if args.gather == "full":
            all_gather(full_all, Y if not ghost else torch.zeros((M, rows), dtype=torch.float64))
        else:
            all_gather(tips_all[0], tips_local[0])
    fence()
    # Kernel duration by HIP events on the launch stream: one pair per launch — except for kernels of a few microseconds
    # (config 2), where an event pair per launch costs more than the kernel: there ONE pair brackets the K launches and the
    # figure is the mean launch-to-launch time, an upper bound of the kernel's duration (rocprofv3's mean is lower).
    probe = [timing_event() for _ in range(2)]
    record(probe[0])
    for i in range(10):
        step(args.warmup + i)
    record(probe[1])
    fence()
    per_launch_events = (not ghost) and elapsed_ms(probe[0], probe[1]) / 10 > 0.1          # ms
    if use_dist:                                             # ranks must agree (ghosts have no clock): any rank's vote
        v = torch.tensor([1.0 if per_launch_events else 0.0], dtype=torch.float64, device=dev)
        all_reduce(v, dist.ReduceOp.MAX)
        per_launch_events = v.item() == 1.0
    # (an event pair costs ~5 us of stream time: bracketing every launch would lengthen every step by 1 %, so every
    # `--event-stride`-th launch is bracketed and the kernel's duration is the mean over those)
    stride = max(1, args.event_stride)
    first = stride // 2                                     # not the first launches of the region: they start from an idle GPU
    events = [(timing_event(), timing_event())
              for _ in range(max(1, (args.steps - first + stride - 1) // stride) if per_launch_events else 1)]
    whole = (timing_event(), timing_event())
    after_gather = timing_event()
    t0 = time.perf_counter()
    record(whole[0])
    if per_launch_events:
        for i in range(args.steps):
            bracket = (i >= first and (i - first) % stride == 0) or (args.steps <= first and i == 0)
            step(i, events[(i - first) // stride] if (bracket and not ghost) else None)
    else:
        events[0] = whole
        for i in range(args.steps):
            step(i)
    record(whole[1])
    if use_dist and args.gather == "final":
        # reassemble the final integrated state of the whole job: one RCCL all-gather over xGMI
        all_gather(tips_all[(args.steps - 1) & 1], tips_local[(args.steps - 1) & 1])
    record(after_gather)
    fence()
    elapsed = time.perf_counter() - t0

    # Two upper bounds of the kernel's duration, both from HIP events on the launch stream over the timed region: the mean
    # bracket of a single launch (execution + the dispatch latency inside the bracket) and the mean launch-to-launch
    # time (execution + the gap between launches; back-to-back launches hide most of the dispatch latency).  The
    # roofline uses the smaller one.
    if ghost:
        bracket_ms = back_to_back_ms = kernel_ms = final_gather_ms = 0.0
    else:
        used = [(a, b) for a, b in events if a is not None]
        bracket_ms = float(np.mean([a.elapsed_time(b) for a, b in used])) / (1 if per_launch_events else args.steps)
        back_to_back_ms = whole[0].elapsed_time(whole[1]) / args.steps
        kernel_ms = min(bracket_ms, back_to_back_ms)
        final_gather_ms = whole[1].elapsed_time(after_gather)
    # max over ranks: the whole timed region (the contract's figure), the kernel, the K steps without the exchange,
    # and the exchange itself (stream time from the last launch's end to the end of the final gather)
    t = torch.tensor([elapsed, kernel_ms, back_to_back_ms, final_gather_ms], dtype=torch.float64, device=dev)
    if use_dist:
        all_reduce(t, dist.ReduceOp.MAX)
    elapsed, kernel_ms, compute_only_ms, final_gather_ms = (float(v) for v in t)
    Xl, Yl = Xs[(args.steps - 1) % nrot], Ys[(args.steps - 1) % nrot]     # the panels of the last step

    # Outside the contract's timed region (and never part of `value`): the same launch, back to back, at least 500 times
    # and for at least ~0.25 s, one HIP-event pair around the block — so that the figure of a 10-ms driver window (K = 20
    # launches of 0.5 ms) is backed by a long run in the same process.  Same panels as the timed steps.
    long_run_ms = long_run_launches = None
    if not ghost and not use_dist and args.long_run > 0:
        est = max(kernel_ms, 1e-3)
        long_run_launches = int(min(max(args.long_run, 250.0 / est), 20000))
        lr = (timing_event(), timing_event())
        record(lr[0])
        for i in range(long_run_launches):
            rc_ = lib.sri_apply(op._h, Xs[i % nrot].data_ptr(), ld, Ys[i % nrot].data_ptr(), ld, M, stream0)
            if rc_ != 0:
                raise sri.SriError(rc_, "sri_apply")
        record(lr[1])
        fence()
        long_run_ms = elapsed_ms(lr[0], lr[1]) / long_run_launches

    # optional: one tip-state gather and one full-panel gather timed on their own (reported, never part of `value`)
    full_gather_ms = None
    tip_gather_ms = None
    if use_dist and args.gather in ("tip", "final"):
        all_gather(tips_all[0], tips_local[0]); fence()
        t1 = time.perf_counter()
        for _ in range(10):
            all_gather(tips_all[0], tips_local[0])
        fence()
        tip_gather_ms = (time.perf_counter() - t1) * 1e3 / 10
        if not rehearsal:                                   # (a rehearsal would stage world x 512 MiB through host memory per rank)
            try:
                buf = torch.empty((M * world, rows), dtype=torch.float64, device=dev)
                all_gather(buf, Yl); fence()
                t1 = time.perf_counter(); all_gather(buf, Yl); fence()
                full_gather_ms = (time.perf_counter() - t1) * 1e3
                del buf
            except Exception as e:                          # noqa: BLE001
                print(f"bench.py: rank {rank}: full-panel gather not measured: {e}", file=sys.stderr)
                full_gather_ms = None

    parity = None
    if rank == 0:
        # post-timing check on 256 sampled columns of the last step's panels (the oracle is the checker, never timed here):
        # the gate of tests/conftest.assert_apply_parity — bit-equality with the CPU model of the kernel's arithmetic (an
        # FMA chain in the handle's k order), Frobenius and backward error against the CPU oracle, and (unless
        # --no-verify: it takes seconds at N = 1024) both against a binary128 product.  A failed gate => EXIT_PARITY.
        import oracle
        A = oracle.dn(n) if which == sri.OP_DN else oracle.integration(n)[0]
        idx = torch.arange(0, M, max(M // 256, 1), device=dev)[:256]
        Xh = np.asfortranarray(Xl[idx].cpu().numpy().T)
        Yh = np.asfortranarray(Yl[idx].cpu().numpy().T)
        asc = op.ascending_k
        forced = os.environ.get("SRI_BENCH_FORCE_MODEL")    # test hook: compare against the WRONG arithmetic model
        if forced in ("paired", "ascending"):
            asc = forced == "ascending"
        limited = args.data == "rod" and which == sri.OP_DN and n >= 1024
        # (--no-verify skips the binary128 product — except where the gate cannot be decided without it)
        g = oracle.parity_gate(A, Xh, Yh, n, ascending=asc, cancellation_limited=limited, exact=(not args.no_verify) or limited)
        if not args.no_verify and "gpu_vs_binary128" not in g:
            Yx, Yr = oracle.apply_exact(A, Xh), oracle.apply(A, Xh, threads=0)
            nx = float(np.linalg.norm(Yx))
            g["gpu_vs_binary128"], g["oracle_vs_binary128"] = float(np.linalg.norm(Yh - Yx) / nx), float(np.linalg.norm(Yr - Yx) / nx)
        parity = {"columns_sampled": int(idx.numel()), "ok": g["ok"], "gate": g["gate"], "failures": g["failures"],
                  "gpu_vs_cpu_oracle_fro": g["fro_vs_oracle"], "backward": g["backward"], "backward_bound": g["backward_bound"],
                  "gpu_vs_binary128_fro": g.get("gpu_vs_binary128"), "cpu_oracle_vs_binary128_fro": g.get("oracle_vs_binary128"),
                  # north_star's own figure, beside whatever tier of the gate applied: Frobenius-relative distance to the CPU oracle <= 1e-12
                  "north_star_tolerance": 1e-12, "north_star_tolerance_met": bool(g["fro_vs_oracle"] <= 1e-12),
                  "k_order": "ascending" if op.ascending_k else "evens before odds in blocks of 8 k",
                  "bit_exact_vs_cpu_model_of_the_kernel": g["bit_exact_vs_fma_model"]}

    gather_ok = None
    slabs_bad = None
    if use_dist and args.gather in ("tip", "final"):
        # every rank's slab of the gathered state must equal that rank's own tips (checked on every rank, reduced) ...
        b_last = (args.steps - 1) & 1
        mine = tips_all[b_last][rank * M:(rank + 1) * M]
        own = tips_local[b_last] if ghost else Yl[:, 0].contiguous()
        ok = torch.tensor([1.0 if torch.equal(mine, own) else 0.0], dtype=torch.float64, device=dev)
        all_reduce(ok, dist.ReduceOp.MIN)
        gather_ok = bool(ok.item() == 1.0)
        if rank == 0:                                       # ... and rank 0 looks at every slab at its offset
            slabs_bad = slab_report(tips_all[b_last], world, M, gpu_ranks, torch)
            gather_ok = gather_ok and not slabs_bad

    rc = EXIT_OK
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rods * world / (elapsed / args.steps)
        flops = 8.0 * rows * rows * rods                    # per launch, algorithmic
        alg_bytes = 64.0 * rows * rods + 8.0 * rows * rows
        mfma_bound = n / 8.0 >= PEAK_FP64_MFMA_TFLOPS * 1e3 / PEAK_HBM_GBS
        # HBM traffic is a PMC measurement (rocprofv3 --pmc, separate passes) that cannot be taken inside this
        # process: the committed summary of tools/profile_bench.sh for the same shape is replayed and labelled.
        traffic, traffic_source, prof_kernel_ms, traffic_stale = None, None, None, None
        sources_sha = lib_sources_sha16()
        ld_tag = "" if ld == rows else f"_ld{ld}"
        for name in (f"hbm_traffic_n{n}_r{rods}_{args.operator}{ld_tag}.json", "hbm_traffic_latest.json"):
            tp = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(tp):
                continue
            try:
                rec = json.load(open(tp))
                if rec.get("n") == n and rec.get("rods") == rods and rec.get("operator") == args.operator and rec.get("ld", rows) == ld:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_stale = rec.get("lib_sources_sha16") != sources_sha
                    traffic_source = f"profiles/{name} (rocprofv3 --pmc passes of tools/profile_bench.sh" \
                                     f"{', ' + rec['date'] if rec.get('date') else ''}; not measured in this run" \
                                     + ("; TAKEN WITH OTHER KERNEL SOURCES than this library's" if traffic_stale else "; same kernel sources") + ")"
                    if rec.get("rocprofv3_kernel_avg_ns") and not traffic_stale:
                        prof_kernel_ms = rec["rocprofv3_kernel_avg_ns"] * 1e-6
                    break
            except Exception:
                traffic = None
        # The roofline is priced with THIS run's clock, always: the smaller of the two HIP-event upper bounds for kernels long
        # enough to bracket, the launch-to-launch mean for kernels of a few microseconds (config 2).  Back-to-back launches of
        # such a kernel OVERLAP (launch i+1 ramps up while launch i drains), so rocprofv3's begin-to-end mean of one dispatch is
        # longer than what a launch costs in a stream: the fraction at THAT duration — from the committed profile, and only if
        # it was taken with this library's sources — is reported beside it (`frac_at_committed_rocprof_duration`), never as `frac`.
        clock = "min(single-launch bracket, launch-to-launch), HIP events in this run" if per_launch_events \
            else "launch-to-launch mean, HIP events in this run"
        priced_ms = kernel_ms
        achieved_tf = flops / (priced_ms * 1e-3) / 1e12
        achieved_gbs = alg_bytes / (priced_ms * 1e-3) / 1e9
        config_id = next((c for c, (cn, cr) in CONFIGS.items() if (cn, cr) == (n, rods) and (c == 4) == (world == 8)), None)
        roof = ({"bound": "mfma", "achieved": achieved_tf, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                 "frac": achieved_tf / PEAK_FP64_MFMA_TFLOPS,
                 "peak_measured": MEASURED_FP64_MFMA_TFLOPS, "frac_of_measured_peak": achieved_tf / MEASURED_FP64_MFMA_TFLOPS}
                if mfma_bound else
                {"bound": "hbm", "achieved": achieved_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved_gbs / PEAK_HBM_GBS,
                 "peak_measured": MEASURED_COPY_GBS, "frac_of_measured_peak": achieved_gbs / MEASURED_COPY_GBS})
        out = {
            "metric": "rod-states/sec (D_N.X apply, fp64) at N=256; % MI355X fp64 MFMA peak" if (n == 256 and which == sri.OP_DN)
                      else f"rod-states/sec ({args.operator} apply, fp64) at N={n}",
            "value": value, "unit": "rod-states/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"N={n}, batch={rods} rods x 4 state columns per GPU, fp64, {args.operator} apply, "
                                   + ("synthetic rod-strain fields" if args.data == "rod" else "uniform U[-1,1) panels")
                                   + (f", {world} GPUs sharded on the batch axis, RCCL all-gather of {args.gather} state" if use_dist else ""),
                       "baseline_config": config_id,
                       "N": n, "rods_per_gpu": rods, "state_columns": 4, "operator": args.operator, "ld": ld,
                       "panel_data": args.data, "panel_pairs_rotated": nrot,
                       "kernel": op.last_kernel, "gather": args.gather if use_dist else "none",
                       "collective": collective,
                       "clock_ramp_ms_untimed": args.ramp_ms,
                       "cus_reserved_for_collective": reserve},
            # AI = N/8 flop/B against the machine balance 78.6 TF / 8 TB/s = 9.8 flop/B: HBM-bound below N ~ 79
            "roofline": roof |
                        {"traffic": traffic, "traffic_source": traffic_source, "traffic_stale": traffic_stale,
                         "lib_sources_sha16": sources_sha, "achieved_TFLOPs": achieved_tf,
                         "kernel_ms": priced_ms, "kernel_ms_clock": clock,
                         "kernel_ms_method": "min of two HIP-event upper bounds over the timed region: "
                                             + ((f"bracket of every {stride}th launch (mean of {len(events)})" if stride > 1 else "bracket of every launch (mean)")
                                                if per_launch_events else "no per-launch brackets (kernel of a few microseconds)")
                                             + " and launch-to-launch mean (one pair around the K launches / K)",
                         "kernel_ms_single_launch_bracket": bracket_ms if per_launch_events else None,
                         "kernel_ms_launch_to_launch": back_to_back_ms,
                         "kernel_ms_rocprofv3_committed": prof_kernel_ms,
                         "frac_at_committed_rocprof_duration": (None if prof_kernel_ms is None else
                                                                (flops / (prof_kernel_ms * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS if mfma_bound
                                                                 else alg_bytes / (prof_kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS)),
                         "kernel_ms_long_run": long_run_ms, "long_run_launches": long_run_launches,
                         "frac_long_run": (None if not long_run_ms else
                                           (flops / (long_run_ms * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS if mfma_bound
                                            else alg_bytes / (long_run_ms * 1e-3) / 1e9 / PEAK_HBM_GBS)),
                         "algorithmic_flops_per_launch": flops,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "hbm_algorithmic_GBps": achieved_gbs,
                         "hbm_frac_of_8TBps": achieved_gbs / PEAK_HBM_GBS},
            "columns_per_s": value * 4,             # the s = 3 position case is this / 3 rod-states/s
            "parity_fro_rel_err_sampled": parity["gpu_vs_cpu_oracle_fro"],
            "parity": parity,
        }
        if use_dist:
            # SURVEY §8(e) (i) compute-only and (ii) end-to-end, side by side: `value` / `ms_per_step` are the contract's
            # whole timed region (K applies + the final gather + the closing barrier, host clock, max over ranks)
            out["ms_per_step_compute_only"] = compute_only_ms
            out["value_compute_only"] = rods * world / (compute_only_ms * 1e-3) if compute_only_ms > 0 else None
            out["final_gather_ms"] = final_gather_ms if args.gather == "final" else None
        if full_gather_ms is not None:
            out["full_panel_allgather_ms"] = full_gather_ms
        if tip_gather_ms is not None:
            out["tip_state_allgather_ms"] = tip_gather_ms
        if gather_ok is not None:
            out["gathered_state_verified"] = gather_ok
            out["gathered_slabs_wrong"] = slabs_bad
        if degraded is not None:
            out["degraded"] = degraded
        if args.cpu_seconds > 0 and not rehearsal:
            # N = 1: the bench contract's CPU baseline.  N > 1: the same object on a quarter of the budget, so that a SCALE record
            # is a strict superset of the BENCH record (rank 0 measures while the other ranks wait at the closing barrier; their
            # waits may spin on cores of the same cgroup, hence `measured_beside_ranks`) — the N = 1 figure is the one to quote.
            out["cpu_baseline"] = cpu_baseline(n, args.cpu_seconds if world == 1 else max(args.cpu_seconds / 4, 2.0), args.data, rods)
            out["cpu_baseline"]["measured_beside_ranks"] = world
        elif args.cpu_seconds > 0:
            out["cpu_baseline"] = {"value": None, "unit": "rod-states/s", "cores": 0, "kind": "port",
                                   "sample": "not measured: rehearsal of a multi-GPU line on one GPU box (ranks share its host cores)"}
        if not parity["ok"]:
            rc = EXIT_PARITY
            print("bench.py: PARITY GATE FAILED: " + "; ".join(parity["failures"]), file=sys.stderr, flush=True)
        elif gather_ok is False:
            rc = EXIT_GATHER
            print(f"bench.py: gathered state wrong (slabs of ranks {slabs_bad})", file=sys.stderr, flush=True)
        elif degraded is not None:
            rc = EXIT_COMM
        out["exit_code"] = rc
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)

    if ghost:
        # self-check of the rehearsal's premise: a ghost rank has never opened the GPU
        fds = []
        for fd in os.listdir("/proc/self/fd"):
            try:
                fds.append(os.readlink(f"/proc/self/fd/{fd}"))
            except OSError:
                pass
        if any(t.startswith("/dev/kfd") or t.startswith("/dev/dri/") for t in fds):
            print(f"bench.py: ghost rank {rank} holds a GPU device node open: {[t for t in fds if t.startswith('/dev/')]}", file=sys.stderr, flush=True)
            rc = 5
    if use_dist:
        # every rank leaves with rank 0's verdict (torchrun reports the first failing rank; all agree here)
        v = torch.tensor([float(rc)], dtype=torch.float64, device=dev)
        all_reduce(v, dist.ReduceOp.MAX)
        rc = int(v.item())
        if degraded is not None and rc == EXIT_OK:
            rc = EXIT_COMM
        barrier()
        if comm_hung:
            # a thread of this rank is stuck inside ncclCommInitRank and cannot be joined or cancelled: interpreter
            # shutdown would wait for RCCL's own teardown.  The line is out and flushed; leave with the failure code.
            print(f"bench.py: rank {rank}: leaving with exit code {rc} past a communicator thread stuck in RCCL", file=sys.stderr, flush=True)
            sys.stdout.flush()
            os._exit(rc)
        if comm_box[0] is not None:
            comm_box[0].close()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    main()
