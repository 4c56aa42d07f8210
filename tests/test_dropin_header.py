"""The drop-in C++ header (include/chebyshev_differentiation.h): same template signatures as the
reference, values bit-identical to the oracle.  Host-only: no GPU needed."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "experimental_gpu_programming_for_a_spectral_numerical_integration_amd")


def _build(tmp_path, src, name, extra=()):
    exe = os.path.join(tmp_path, name)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-L", PKG, "-lsri_hip",
           f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe, *extra]
    subprocess.check_call(cmd)
    return exe


def test_reference_signatures_compile_and_match_oracle(sri, oracle, tmp_path):
    exe = _build(str(tmp_path), os.path.join(ROOT, "tests", "cpp", "dropin_check.cpp"), "dropin_check")
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split("\n")
    it = iter(out)
    for n in (2, 16, 64):
        assert next(it) == f"N {n}"
        x, x3, c = oracle.points(n), oracle.points(n, 3.0), oracle.coeffs(n)
        for i in range(n):
            _, a, b, _, cc = next(it).split()
            assert float(a) == x[i] and float(b) == x3[i] and float(cc) == c[i]
        D = oracle.dn(n)
        got = np.array([float(next(it).split()[1]) for _ in range(n * n)]).reshape(n, n).T   # column-major dump
        assert np.array_equal(got, D)
        _, a, b = next(it).split()
        assert float(a) == D[n - 2, n - 2] and float(b) == D[0, n - 1]
    tag, rows, d00 = next(it).split()
    assert tag == "big" and int(rows) == 300 and float(d00) == oracle.dn(300)[0, 0]


def test_header_has_the_reference_signatures():
    text = open(os.path.join(ROOT, "include", "chebyshev_differentiation.h")).read()
    for sig in ("template <unsigned int t_number_of_chebyshev_nodes, unsigned int t_L = 1>",
                "static std::array<double, t_number_of_chebyshev_nodes> ComputeChebyshevPoints()",
                "static std::array<double, t_number_of_chebyshev_nodes> GetCoefficients_c()",
                "getDn()"):
        assert sig in text
    assert "#ifndef CHEBYSHEV_DIFFERENTIATION_H" in text            # same include guard as the reference header


def test_cpp_multi_gpu_example_compiles(sri, tmp_path):
    """The C++-only multi-GPU example builds against the C ABI (run on the GPU box by tests/test_examples_gpu.py)."""
    _build(str(tmp_path), os.path.join(ROOT, "examples", "multi_gpu_gather.cpp"), "multi_gpu_gather",
           extra=("-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L/opt/rocm/lib", "-lamdhip64", "-pthread", "-Werror"))
