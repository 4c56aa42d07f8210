"""BASELINE config 1 through the product: examples/rod_main.cpp (main.cpp restated on the drop-in
header, position integration on the GPU) against the oracle's restatement of main.cpp and against the stdout of the
reference's own main.cpp (tests/golden/config1_stdout.txt)."""
import os
import subprocess

import numpy as np
import pytest

from test_dropin_header import ROOT, _build

pytestmark = pytest.mark.gpu


def _parse(out):
    lines = out.strip().split("\n")
    qi, ri = lines.index("Q_stack : "), lines.index("r_stack : ")
    Q = np.array([float(v) for v in lines[qi + 1:ri]])
    r = np.array([[float(v) for v in ln.split()] for ln in lines[ri + 1:ri + 16]])
    return Q, r


def test_rod_main_config1_matches_oracle(sri, oracle, tmp_path):
    exe = _build(str(tmp_path), os.path.join(ROOT, "examples", "rod_main.cpp"), "rod_main")
    out = subprocess.run([exe, "--full"], capture_output=True, text=True, check=True).stdout
    Q, r = _parse(out)
    Qo = oracle.integrate_quaternions(16, oracle.QE_MAIN_CPP)          # (15, 4) column-major
    ro = oracle.integrate_position(16, Qo)
    assert np.abs(Q - Qo.T.reshape(-1)).max() <= 1e-14                 # elimination vs explicit inverse: rounding only
    assert np.abs(r - ro).max() <= 1e-13
    # default formatting = what main.cpp:198,202 print through Eigen's operator<<.  The f4 fixture
    # tests/golden/config1_stdout.txt is the stdout of the reference's own, unmodified main.cpp built against this repo
    # (tests/test_reference_main.py); the example must print the same text.  Only coefficients that are rounding noise
    # around an exact zero (|v| < 1e-12: two LU variants legitimately differ there) may differ as text.
    out6 = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    want = open(os.path.join(ROOT, "tests", "golden", "config1_stdout.txt")).read()
    got_lines, want_lines = out6.split("\n"), want.split("\n")
    assert len(got_lines) == len(want_lines)
    for g, w in zip(got_lines, want_lines):
        if g == w:
            continue
        gt, wt = g.split(), w.split()
        assert len(gt) == len(wt), (g, w)
        for a, b in zip(gt, wt):
            assert a == b or (abs(float(a)) < 1e-12 and abs(float(b)) < 1e-12), (g, w)
    assert "0.79977\n" in out6 and "0.800067\n" in out6
    # batched: 100 rods through one device product
    outb = subprocess.run([exe, "100", "--full"], capture_output=True, text=True, check=True).stdout
    Qb, rb = _parse(outb)
    assert np.abs(rb - ro).max() <= 1e-13 and "sri_apply_mfma_f64" in outb


@pytest.mark.parametrize("force_ragged", ["0", "1"])
def test_multi_gpu_gather_from_cpp(sri, tmp_path, force_ragged):
    """examples/multi_gpu_gather.cpp: the sharded apply + RCCL all-gather of the tip state from C++ only (host threads, one
    per GPU; the box has one, so the communicator has one rank — the same ncclCommInitRank / all-gather calls)."""
    exe = _build(str(tmp_path), os.path.join(ROOT, "examples", "multi_gpu_gather.cpp"), "multi_gpu_gather",
                 extra=("-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L/opt/rocm/lib", "-lamdhip64", "-pthread"))
    env = dict(os.environ, SRI_DEBUG="comm_force_ragged=" + force_ragged, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, "1", "1003"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "verified" in r.stdout, (r.stdout, r.stderr[-2000:])
