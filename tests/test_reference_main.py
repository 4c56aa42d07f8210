"""the reference's OWN driver, /root/reference/main.cpp (+ include/utilities.h), compiled in place
and UNMODIFIED against this repo's drop-in header (include/), the product library and the self-written compatibility
headers under compat/ (Eigen subset, Boost legendre_p) — no Eigen, no Boost on the machine.  Host formers only, so it
runs without a GPU.  Its stdout is BASELINE config 1 (N=16, one rod, the hard-coded qe of main.cpp:187-195).

This is product-side evidence (the drop-in claim, end to end); it is not an oracle pin: the binary's arithmetic is this
repo's (our D_N former, our stand-in inverse / product), only the driver source is the reference's.  Nothing from
/root/reference is copied: the test is skipped where the tree is absent (the GPU box)."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "experimental_gpu_programming_for_a_spectral_numerical_integration_amd")
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")

needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "main.cpp")),
                                     reason="/root/reference is not present on this machine")


def build_reference_main(out_dir, extra=()):
    exe = os.path.join(str(out_dir), "reference_main")
    cmd = ["g++", "-std=c++20", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "compat"),
           "-I", os.path.join(REF, "include"), os.path.join(REF, "main.cpp"), "-L", PKG, "-lsri_hip",
           f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe, *extra]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return exe


def parse_stdout(text):
    lines = text.split("\n")
    qi, ri = lines.index("Q_stack : "), lines.index("r_stack : ")
    Q = np.array([float(v) for v in lines[qi + 1:ri]])
    r = np.array([[float(v) for v in ln.split()] for ln in lines[ri + 1:] if ln.strip()])
    return Q, r


@needs_reference
def test_unmodified_reference_main_builds_and_reproduces_config1(sri, tmp_path):
    # our header must be the one main.cpp's `#include "chebyshev_differentiation.h"` finds: -I include/ comes first and
    # the reference keeps its copy under include/, not next to main.cpp
    assert not os.path.exists(os.path.join(REF, "chebyshev_differentiation.h"))
    exe = build_reference_main(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    Q, r = parse_stdout(out)
    assert Q.shape == (60,) and r.shape == (15, 3)
    anchors = json.load(open(os.path.join(GOLD, "config1_anchors.json")))
    for node in anchors["nodes"]:
        i = node["i"]
        # std::cout prints 6 significant digits (main.cpp:198,202)
        assert abs(Q[i] - node["q_w"]) <= 5.1e-6 * abs(node["q_w"]) and abs(Q[30 + i] - node["q_y"]) <= 5.1e-6 * abs(node["q_y"])
        assert Q[15 + i] == 0 and Q[45 + i] == 0 and r[i, 1] == 0
        assert abs(r[i, 0] - node["r_x"]) <= 5.1e-6 * abs(node["r_x"]) and abs(r[i, 2] - node["r_z"]) <= 5.1e-6 * abs(node["r_z"])
    # the committed f4 fixture IS this program's stdout
    assert out == open(os.path.join(GOLD, "config1_stdout.txt")).read()


@needs_reference
def test_reference_main_under_sanitizers(sri, tmp_path):
    """The stand-in headers under AddressSanitizer + UBSan + libstdc++ assertions (CPU build only)."""
    exe = build_reference_main(tmp_path, extra=("-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                                                "-D_GLIBCXX_ASSERTIONS"))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([exe], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout == open(os.path.join(GOLD, "config1_stdout.txt")).read()


def test_stdout_fixture_agrees_with_the_oracle(oracle):
    """The f4 fixture against the CPU oracle's restatement of main.cpp, at the 6 significant digits it holds."""
    Q, r = parse_stdout(open(os.path.join(GOLD, "config1_stdout.txt")).read())
    Qo = oracle.integrate_quaternions(16, oracle.QE_MAIN_CPP)          # (15, 4)
    ro = oracle.integrate_position(16, Qo)
    fmt = np.vectorize(lambda v: float("%.6g" % v))
    clean = lambda a: np.where(np.abs(a) < 1e-15, 0.0, a)
    assert np.array_equal(Q, fmt(clean(Qo.T.reshape(-1))))
    assert np.array_equal(r, fmt(clean(ro)))


def test_header_alone_provides_what_main_cpp_needs(sri, tmp_path):
    """include/chebyshev_differentiation.h as the ONLY include: std::array, std::cout, <cmath>, <algorithm> must come with
    it, as they do with the reference header (h:10-11) — with the compat Eigen and without any Eigen at all."""
    src = os.path.join(ROOT, "tests", "cpp", "header_only_tu.cpp")
    for extra in ([], ["-I", os.path.join(ROOT, "compat")]):
        exe = os.path.join(str(tmp_path), "header_only_tu")
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), *extra, src,
                               "-L", PKG, "-lsri_hip", f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
        assert out == "x0 : \n1 2 16\n"


def test_compat_eigen_subset(tmp_path):
    """Unit checks of the stand-in headers on their own (no library needed)."""
    src = os.path.join(ROOT, "tests", "cpp", "compat_eigen_check.cpp")
    exe = os.path.join(str(tmp_path), "compat_eigen_check")
    subprocess.check_call(["g++", "-std=c++20", "-Wall", "-Wextra", "-Werror", "-g", "-fsanitize=address,undefined",
                           "-D_GLIBCXX_ASSERTIONS", "-I", os.path.join(ROOT, "compat"), src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and r.stdout.strip().endswith("compat ok"), (r.stdout[-2000:], r.stderr[-2000:])
