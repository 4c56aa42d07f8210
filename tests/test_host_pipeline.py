"""The host-panel pipeline of sri_apply_host / sri_integrate_host (csrc/sri_host_pipeline.h) on the CPU, against a mock card.

The driver that orders IN / UP / RUN / DOWN / OUT over the staging ring and the copy threads is pure C++ and a template over
the device side; tests/cpp/host_pipeline_check.cpp instantiates it with a "card" made of threads (three in-order streams,
per-slot events, random delays, results that depend on the global column index, staging slots scribbled while the card owns
them) and checks every value — plain, and under ThreadSanitizer, where a staging slot touched by the driver and the card
without an ordering between them is a reported race even if the values happen to survive.
The real card (HIP streams, pinned slots) is tests/test_host_panels_gpu.py's subject."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "experimental_gpu_programming_for_a_spectral_numerical_integration_amd", "csrc")
SRC = os.path.join(ROOT, "tests", "cpp", "host_pipeline_check.cpp")


def _build(tmp_path, name, flags):
    exe = os.path.join(str(tmp_path), name)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-pthread", "-I", CSRC, *flags, SRC, "-o", exe])
    return exe


def test_pipeline_orders_every_stage_against_a_mock_card(tmp_path):
    exe = _build(tmp_path, "host_pipeline_check", ["-O2"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "host_pipeline_check: 0 failures" in out.stdout


def test_pipeline_is_race_free_under_thread_sanitizer(tmp_path):
    exe = _build(tmp_path, "host_pipeline_check_tsan", ["-O1", "-g", "-fsanitize=thread"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1")
    out = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "ThreadSanitizer" not in out.stderr, out.stdout[-3000:] + out.stderr[-6000:]
    assert "host_pipeline_check: 0 failures" in out.stdout


def test_the_library_never_page_locks_or_registers_caller_memory():
    """By construction (DESIGN.md §4 "Host panels"): no hipHostRegister / hipHostUnregister anywhere in the product, and the
    host-panel translation unit hands HIP only its own pinned slots (or memory the caller declared page-locked)."""
    for name in os.listdir(CSRC):
        if not name.endswith((".hip", ".cpp", ".h")):
            continue
        text = open(os.path.join(CSRC, name)).read()
        code = "\n".join(line.split("//")[0] for line in text.splitlines())
        assert "hipHostRegister(" not in code and "hipHostUnregister(" not in code, name
    import subprocess as sp
    lib = os.path.join(os.path.dirname(CSRC), "libsri_hip.so")
    if os.path.exists(lib):
        undefined = sp.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True).stdout
        assert "hipHostRegister" not in undefined and "hipHostUnregister" not in undefined
