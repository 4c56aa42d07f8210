"""The C ABI's RCCL communicator (csrc/sri_comm.hip) on the one-GPU box: a 1-rank communicator runs the same
ncclCommInitRank / ncclAllGather / grouped-broadcast calls an 8-rank job makes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r"""
import sys, torch
sys.path.insert(0, {root!r})
import experimental_gpu_programming_for_a_spectral_numerical_integration_amd as sri
from experimental_gpu_programming_for_a_spectral_numerical_integration_amd import parallel
torch.cuda.set_device(0)
uid = parallel.Communicator.unique_id()
assert len(uid) == 128
comm = parallel.Communicator(uid, 0, 1, 0)
n, rods = 64, 37
op = sri.SpectralApply(n, sri.OP_DN, device=0)
X = torch.rand((4 * rods, n), dtype=torch.float64, device="cuda") * 2 - 1
tips = torch.empty(4 * rods, dtype=torch.float64, device="cuda")
Y = op.apply(X, tips=tips)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):                       # the collective runs on the caller's stream
    all_tips = comm.allgather_tips(tips)
    all_tips_r = comm.allgather_tips(tips, columns_per_rank=[4 * rods])
    full = comm.allgather_panel(Y)
    full_r = comm.allgather_panel(Y, columns_per_rank=[4 * rods])
side.synchronize()
assert torch.equal(all_tips, Y[:, 0].contiguous()) and torch.equal(all_tips_r, all_tips)
assert torch.equal(full, Y) and torch.equal(full_r, Y)
t2, _ = parallel.gather_tip_state(Y, comm=comm)
p2, _ = parallel.gather_panel(Y, comm=comm)
torch.cuda.synchronize()
assert torch.equal(t2, all_tips) and torch.equal(p2, Y)
try:
    comm.allgather_tips(tips, columns_per_rank=[5])
    raise SystemExit("mismatched counts accepted")
except ValueError:
    pass
comm.close()
print("COMM_OK")
"""


@pytest.mark.parametrize("force_ragged", ["0", "1"])
def test_one_rank_communicator(force_ragged):
    env = dict(os.environ, SRI_COMM_FORCE_RAGGED=force_ragged, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", BODY.format(root=ROOT)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "COMM_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
