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


def _run(cmd, env=None):
    e = dict(os.environ, **(env or {}))
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, nothing else on stdout
    return json.loads(lines[0])


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


def test_one_rank_rccl_group_runs_the_multi_gpu_path():
    for mode in ("final", "tip"):
        d = _run([sys.executable, "bench.py", "--gather", mode] + SMALL, env={"SRI_BENCH_FORCE_DIST": "1"})
        assert d["config"]["gather"] == mode and d["gathered_state_verified"] is True
        assert d["tip_state_allgather_ms"] > 0


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
