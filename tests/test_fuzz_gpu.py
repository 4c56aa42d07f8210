"""A short run of tools/fuzz_apply.py (randomised operators, shapes, strides, offsets, poisoned padding; bit-exact against
the CPU model of the kernel's arithmetic; apply, tips, integrate, host panels).  The long run is in profiles/."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4])
def test_fuzz_apply(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_apply.py"), "--cases", "120", "--seed", str(seed)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FUZZ CLEAN" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
