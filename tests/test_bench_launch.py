"""bench.py's own rank launcher and presets, without a GPU: `python bench.py --gpus N` (N > 1, no torchrun
environment) must start torch.distributed.run as a CHILD process, relay rank 0's one JSON line and the return code."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_SCRIPT = textwrap.dedent("""
    import json, os, sys
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1"
    print("library banner on stdout")                 # chatter must not reach the parent's stdout
    if "--fail" in sys.argv and rank == 1:
        sys.exit(3)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "argv": sys.argv[1:]}))
""")


def _launch(tmp_path, argv):
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    code = f"import sys; sys.path.insert(0, {ROOT!r}); import bench; sys.exit(bench.launch_ranks(2, {argv!r}, script={str(script)!r}))"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)


def test_launcher_relays_one_line_and_rc(tmp_path):
    r = _launch(tmp_path, ["--gpus", "2", "--steps", "5"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["argv"] == ["--gpus", "2", "--steps", "5"]
    assert "library banner" in r.stderr


def test_launcher_propagates_failure(tmp_path):
    r = _launch(tmp_path, ["--fail"])
    assert r.returncode != 0


def test_main_routes_unwrapped_multi_gpu_through_the_launcher_before_torch(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    seen = {}

    def fake(n, argv, script=None):
        seen["n"], seen["argv"] = n, list(argv)
        return 0

    monkeypatch.setattr(bench, "launch_ranks", fake)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert seen["n"] == 4 and seen["argv"] == ["--gpus", "4", "--steps", "7"]


def test_config_presets_match_baseline_json():
    sys.path.insert(0, ROOT)
    import bench

    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    for cfg, (n, rods) in bench.CONFIGS.items():
        text = "".join(base[cfg - 1].split())                # "65 536" -> "65536"
        assert f"N={n}," in text
        assert f"batch={rods * (8 if cfg == 4 else 1)}" in text
    a = bench.parse(["--config", "2"]); assert (a.n, a.rods) == (64, 4096)
    a = bench.parse([]); assert (a.n, a.rods, a.data) == (256, 65536, "rod")
    a = bench.parse(["--config", "5", "--rods", "100"]); assert (a.n, a.rods) == (1024, 100)


def test_host_and_device_panels_are_the_same_field():
    import numpy as np
    import torch

    sys.path.insert(0, ROOT)
    import bench

    X = bench.device_panel(torch, 32, 32, 5, "rod", 7, "cpu")          # (4*rods, rows)
    assert X.shape == (20, 32)
    x = X.reshape(5, 4, 32).numpy()
    # columns 0..2 are quadratics in the node coordinate (exact degree-2 fit), column 3 = cos|K|
    t = np.cos(np.pi * np.arange(32) / 31)
    for r in range(5):
        for c in range(3):
            coef = np.polyfit(t, x[r, c], 2)
            assert np.max(np.abs(np.polyval(coef, t) - x[r, c])) < 1e-12
        assert np.allclose(x[r, 3], np.cos(np.sqrt((x[r, :3] ** 2).sum(axis=0))), atol=1e-14)
    H = bench.host_panel(32, 5, "rod", 7)
    assert H.shape == (32, 20) and H.flags.f_contiguous


def test_eigen_reference_library_baseline_is_probed_not_faked():
    """SURVEY 8(d) CPU plan item 3: bench.py times Eigen's own A*X where a REAL Eigen is installed, and says so where it
    is not (this image) — it must never build against this repo's compat/ stand-ins."""
    sys.path.insert(0, ROOT)
    import bench

    r = bench.eigen_baseline(64, 64)
    assert r["available"] in (True, False)
    if not r["available"]:
        assert "searched" in r or "error" in r
    os.environ["EIGEN3_INCLUDE_DIR"] = os.path.join(ROOT, "compat")        # pointing it at the stand-ins must not count
    try:
        r2 = bench.eigen_baseline(64, 64)
        assert r2["available"] == r["available"]
    finally:
        del os.environ["EIGEN3_INCLUDE_DIR"]
