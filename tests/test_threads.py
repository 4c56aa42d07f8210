"""Thread safety of the C ABI's handle creation: a ThreadSanitizer HOST build of the library (hipcc
applies -fsanitize=thread to the host side only) driven from two threads.  On this GPU-less box every create must
return SRI_ENODEVICE from both threads, with no race report; tools/ can run the same binary on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "experimental_gpu_programming_for_a_spectral_numerical_integration_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build_tsan(out_dir):
    lib = os.path.join(out_dir, "libsri_hip_tsan.so")
    srcs = ["sri_host.cpp", "sri_apply.hip", "sri_apply_small.hip", "sri_rod.hip", "sri_rod_resident.hip", "sri_comm.hip"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-fsanitize=thread", "-Wno-option-ignored", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                           "-I", "/opt/rocm/include", "-shared", "-o", lib, "-x", "hip", *[os.path.join(CSRC, s) for s in srcs],
                           "-ldl"], stderr=subprocess.DEVNULL)
    exe = os.path.join(out_dir, "threads_check")
    subprocess.check_call([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "threads_check.cpp"), "-L", out_dir, "-lsri_hip_tsan",
                           f"-Wl,-rpath,{out_dir}", "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", exe])
    return exe


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang not present")
def test_two_threads_create_handles_without_a_race(tmp_path):
    exe = build_tsan(str(tmp_path))
    for mode in ("different", "same"):
        r = subprocess.run([exe, mode], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
        assert r.returncode == 0, (mode, r.stdout[-500:], r.stderr[-3000:])
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
        fields = r.stdout.split()
        assert fields[3:5] == ["0", "0"]                      # both workers ran to the end
        import torch
        if not torch.cuda.is_available():
            assert fields[0:2] == ["devices", "0"] and fields[6:8] == ["0", "0"] and fields[9:11] == ["18", "18"], r.stdout
