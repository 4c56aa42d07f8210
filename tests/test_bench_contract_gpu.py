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
