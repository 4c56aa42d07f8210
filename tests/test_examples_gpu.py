"""BASELINE config 1 through the product: examples/rod_main.cpp (main.cpp restated on the drop-in
header, position integration on the GPU) against the oracle's restatement of main.cpp."""
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
    # default formatting = what main.cpp:198,202 print through Eigen's operator<<: "%.6g", right-aligned to the
    # widest coefficient, one space between columns (textual diff against the reference)
    out6 = subprocess.run([exe], capture_output=True, text=True, check=True).stdout

    def eigen_like(M):
        txt = [["%.6g" % v for v in row] for row in M]
        w = max(len(t) for row in txt for t in row)
        return "\n".join(" ".join(t.rjust(w) for t in row) for row in txt) + "\n"

    # formatted from the example's own full-precision values (entries at rounding-noise level, 1e-17 vs an exact
    # 0, legitimately differ between two LU variants and cannot be compared as text)
    expect = "Q_stack : \n" + eigen_like(Q.reshape(-1, 1)) + "r_stack : \n" + eigen_like(r)
    assert out6 == expect
    assert "0.79977\n" in out6 and "0.800067\n" in out6
    # batched: 100 rods through one device product
    outb = subprocess.run([exe, "100", "--full"], capture_output=True, text=True, check=True).stdout
    Qb, rb = _parse(outb)
    assert np.abs(rb - ro).max() <= 1e-13 and "sri_apply_mfma_f64" in outb
