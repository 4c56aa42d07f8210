"""bench.py as the driver runs it: exactly one JSON line on stdout with the contract's keys, for N=1, for a
1-rank RCCL group (the N > 1 code path on one GPU) and for a 2-rank rehearsal launched by torch.distributed.run
(all ranks share cuda:0, collectives staged over gloo)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "20", "--warmup", "3", "--ramp-ms", "200", "--rods", "4096", "--cpu-seconds", "1"]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _run(cmd, env=None, rc=0, drop=()):
    e = {k: v for k, v in dict(os.environ, **(env or {})).items() if k not in drop}
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == rc, (r.returncode, r.stderr[-2000:])
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, nothing else on stdout
    d = json.loads(lines[0])
    assert d["exit_code"] == rc
    d["_stderr"] = r.stderr
    return d


def test_single_gpu_line():
    d = _run([sys.executable, "bench.py"] + SMALL)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3
    assert d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert d["unit"] == "rod-states/s" and d["value"] > 1e6
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0
    assert d["parity_fro_rel_err_sampled"] <= 1e-12
    assert "workload" in d["config"] and "model" not in d["config"]
    # measurement hygiene (round 4): `frac` is priced with THIS run's clock; a committed PMC summary is replayed only as
    # labelled side information, keyed on the hash of the library's sources; a long back-to-back run backs the short window
    assert r["kernel_ms"] == min(x for x in (r["kernel_ms_single_launch_bracket"], r["kernel_ms_launch_to_launch"]) if x)
    assert len(r["lib_sources_sha16"]) == 16 and r["long_run_launches"] >= 500
    assert abs(r["kernel_ms_long_run"] / r["kernel_ms"] - 1) < 0.1 and abs(r["frac_long_run"] / r["frac"] - 1) < 0.1
    assert r["traffic"] is None or r["traffic_stale"] in (True, False)


def test_small_kernel_roofline_is_priced_with_this_runs_clock():
    d = _run([sys.executable, "bench.py", "--config", "2", "--steps", "200", "--warmup", "20", "--ramp-ms", "200", "--cpu-seconds", "0"])
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["kernel_ms"] == r["kernel_ms_launch_to_launch"] and "this run" in r["kernel_ms_clock"]
    assert abs(r["frac"] - r["hbm_algorithmic_GBps"] / 8000.0) < 1e-12
    if r["kernel_ms_rocprofv3_committed"] is not None:      # only ever replayed for the same kernel sources
        assert r["traffic_stale"] is False and r["frac_at_committed_rocprof_duration"] > 0


def test_one_rank_rccl_group_runs_the_multi_gpu_path():
    for mode in ("final", "tip"):
        d = _run([sys.executable, "bench.py", "--gather", mode] + SMALL, env={"SRI_BENCH_FORCE_DIST": "1"})
        assert d["config"]["gather"] == mode and d["gathered_state_verified"] is True
        assert d["tip_state_allgather_ms"] > 0
        # the multi-GPU code path carries everything the N = 1 line carries (round 5): a real SCALE record can be checked
        # against a BENCH record key by key — roofline, the CPU baseline measured on rank 0, compute-only and gather figures
        assert d["roofline"]["frac"] > 0 and d["roofline"]["kernel_ms"] > 0
        cb = d["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["measured_beside_ranks"] == 1
        assert d["ms_per_step_compute_only"] > 0 and d["value_compute_only"] >= d["value"] and "final_gather_ms" in d
        assert d["parity"]["north_star_tolerance_met"] is True


def test_two_rank_rehearsal_under_torchrun():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", "bench.py", "--gpus", "2"] + SMALL
    d = _run(cmd, env={"SRI_BENCH_REHEARSAL": "1"})
    assert d["n_gpus"] == 2 and d["gathered_state_verified"] is True
    assert "2 GPUs" in d["config"]["workload"]


def test_unwrapped_multi_gpu_invocation_launches_its_own_ranks():
    """The driver's form: plain `python bench.py --gpus 2 ...` (no torchrun wrapper).  bench.py must start the ranks
    itself as a child process and relay one line."""
    env = {"SRI_BENCH_REHEARSAL": "1"}
    clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + SMALL, cwd=ROOT, env=dict(clean, **env),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["steps"] == 20 and d["gathered_state_verified"] is True
    assert "2 GPUs" in d["config"]["workload"]


def test_one_rank_group_uses_the_c_abi_communicator():
    d = _run([sys.executable, "bench.py", "--gather", "final"] + SMALL, env={"SRI_BENCH_FORCE_DIST": "1"})
    assert d["config"]["collective"].startswith("sri_comm"), d["config"]["collective"]
    d = _run([sys.executable, "bench.py", "--gather", "full", "--collective", "torch"] + SMALL, env={"SRI_BENCH_FORCE_DIST": "1"})
    assert d["config"]["collective"].startswith("torch.distributed")


def test_config_presets_and_data_kinds():
    d = _run([sys.executable, "bench.py", "--config", "2", "--steps", "50", "--warmup", "5", "--ramp-ms", "100", "--cpu-seconds", "1"])
    assert d["config"]["N"] == 64 and d["config"]["rods_per_gpu"] == 4096 and d["config"]["baseline_config"] == 2
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["unit"] == "GB/s"
    assert d["config"]["panel_data"] == "rod" and d["parity_fro_rel_err_sampled"] <= 1e-12
    cb = d["cpu_baseline"]
    assert cb["reps"] >= 3 and cb["median_value"] <= cb["value"] and cb["all_cores_median_value"] <= cb["all_cores_value"]
    d = _run([sys.executable, "bench.py", "--data", "uniform"] + SMALL)
    assert d["config"]["panel_data"] == "uniform" and "uniform" in d["config"]["workload"]


TORCHRUN_ENV = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")


def test_config4_eight_rank_rehearsal_at_the_real_per_rank_shape():
    """BASELINE config 4 — the one configuration this pool cannot run — exactly as the driver will invoke it, on the one
    device: 8 ranks x 65 536 rods per rank (the real per-rank shape; the first ranks run the apply on cuda:0, the rest are
    GPU-less ghost ranks because a box admits six processes on its card), collectives staged over gloo.  The line must
    say 8 GPUs / config 4, and every rank's slab of the gathered final state must sit at its own offset."""
    d = _run([sys.executable, "bench.py", "--gpus", "8", "--config", "4", "--steps", "5", "--warmup", "2"],
             env={"SRI_BENCH_REHEARSAL": "1", "SRI_BENCH_REHEARSAL_GPU_RANKS": "4"}, drop=TORCHRUN_ENV)
    assert KEYS <= set(d) and d["n_gpus"] == 8 and d["steps"] == 5 and d["warmup"] == 2
    c = d["config"]
    assert c["baseline_config"] == 4 and c["N"] == 256 and c["rods_per_gpu"] == 65536 and c["gather"] == "final"
    assert "8 GPUs" in c["workload"] and "ghost" in c["collective"]
    assert d["gathered_state_verified"] is True and d["gathered_slabs_wrong"] == []
    assert d["parity"]["ok"] is True and d["parity"]["gate"] == "fro<=1e-12" and d["parity"]["bit_exact_vs_cpu_model_of_the_kernel"]
    # compute-only and end-to-end figures side by side (SURVEY 8(e) i / ii)
    assert d["ms_per_step_compute_only"] > 0 and d["final_gather_ms"] >= 0 and d["value_compute_only"] >= d["value"]
    # The N > 1 line is a strict superset of the N = 1 line — top level, `roofline`, `parity`, `config`, `cpu_baseline` — so that
    # the first real SCALE record can be checked against a BENCH record key by key.  (In the rehearsal the CPU baseline is a
    # labelled placeholder: eight ranks share one box's host cores.)
    one = _run([sys.executable, "bench.py"] + SMALL)
    one.pop("_stderr"); d.pop("_stderr")
    assert set(one) <= set(d), set(one) - set(d)
    for obj in ("roofline", "parity", "config", "cpu_baseline"):
        assert set(one[obj]) >= {"value", "unit", "cores", "kind", "sample"} if obj == "cpu_baseline" else True
        missing = set(one[obj]) - set(d[obj]) if obj != "cpu_baseline" else {"value", "unit", "cores", "kind", "sample"} - set(d[obj])
        assert not missing, (obj, missing)
    assert d["parity"]["north_star_tolerance_met"] is True and one["parity"]["north_star_tolerance_met"] is True


def test_single_gpu_line_is_a_superset_of_round_2s():
    d = _run([sys.executable, "bench.py"] + SMALL)
    r2 = {"traffic", "traffic_source", "achieved_TFLOPs", "kernel_ms", "kernel_ms_method", "kernel_ms_single_launch_bracket",
          "kernel_ms_launch_to_launch", "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch", "hbm_algorithmic_GBps",
          "hbm_frac_of_8TBps", "bound", "achieved", "peak", "unit", "frac"}
    assert r2 <= set(d["roofline"])
    assert abs(d["roofline"]["frac_of_measured_peak"] - d["roofline"]["achieved"] / 77.5) < 1e-12
    assert {"columns_sampled", "gpu_vs_cpu_oracle_fro", "gpu_vs_binary128_fro", "cpu_oracle_vs_binary128_fro", "k_order",
            "bit_exact_vs_cpu_model_of_the_kernel", "gate", "backward", "ok"} <= set(d["parity"])
    assert "ms_per_step_compute_only" not in d and "degraded" not in d


@pytest.mark.parametrize("inject", ["comm_fail", "comm_hang", "comm_probe_fail"])
def test_a_broken_communicator_never_ends_in_rc_0(inject):
    """sri_comm_create failing, hanging, or RCCL not loadable on a rank: the measurement is still printed (from torch's
    RCCL, marked degraded), the rank and its error are named on stderr, and the exit code is 3."""
    d = _run([sys.executable, "bench.py", "--gather", "final"] + SMALL,
             env={"SRI_BENCH_FORCE_DIST": "1", "SRI_BENCH_INJECT": inject, "SRI_COMM_TIMEOUT_S": "3"}, rc=3)
    assert "sri_comm unavailable" in d["degraded"] and "rank 0" in d["degraded"]
    assert "DEGRADED" in d["config"]["collective"] and d["gathered_state_verified"] is True
    assert "rank 0: sri_comm unavailable" in d["_stderr"]


def test_bench_exits_nonzero_when_the_parity_gate_fails():
    """(1) the right kernel checked against the WRONG arithmetic model; (2) a deliberately broken build of the library
    (one operator entry off by 2^-20 relative: `make -C csrc broken`) checked against the right one.  Both: rc 2."""
    d = _run([sys.executable, "bench.py"] + SMALL, env={"SRI_BENCH_FORCE_MODEL": "ascending"}, rc=2)
    assert d["parity"]["ok"] is False and d["parity"]["bit_exact_vs_cpu_model_of_the_kernel"] is False
    assert "PARITY GATE FAILED" in d["_stderr"]
    csrc = os.path.join(ROOT, "experimental_gpu_programming_for_a_spectral_numerical_integration_amd", "csrc")
    subprocess.check_call(["make", "-j4", "-C", csrc, "broken"], stdout=subprocess.DEVNULL)
    d = _run([sys.executable, "bench.py", "--no-verify"] + SMALL,
             env={"SRI_HIP_LIBRARY": os.path.join(ROOT, "build", "libsri_hip_broken.so")}, rc=2)
    assert d["parity"]["ok"] is False and d["parity"]["failures"]
    # ... and the one-tile kernel a small batch of a 64-row operator takes has its own packing: wrong there too
    d = _run([sys.executable, "bench.py", "--no-verify", "--n", "64"] + SMALL,
             env={"SRI_HIP_LIBRARY": os.path.join(ROOT, "build", "libsri_hip_broken.so")}, rc=2)
    assert d["parity"]["ok"] is False and "one-tile-per-workgroup" in d["config"]["kernel"]
