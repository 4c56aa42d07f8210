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
    # default formatting = 6 significant digits like Eigen's operator<< (main.cpp:198,202)
    out6 = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    assert out6.startswith("Q_stack : \n0.79977\n")
    # batched: 100 rods through one device product
    outb = subprocess.run([exe, "100", "--full"], capture_output=True, text=True, check=True).stdout
    Qb, rb = _parse(outb)
    assert np.abs(rb - ro).max() <= 1e-13 and "sri_apply_mfma_f64" in outb
