"""dffw_comm_* / dffw_allgather: the RCCL all-gather of the C ABI (include/dffw.h) that collects the
per-rank depth maps of a batch sharded over the GPUs of one node — what nn.DataParallel's gather does in the reference
(Depth_Estimation_Test/test.py:32).  On a one-GPU box only the single-rank communicator can run; the two-rank test spawns
one process per GPU and is skipped below two GPUs."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_allgather_single_rank_is_a_copy(lib_built):
    from dffinthewild_amd import engine as eng
    lib = eng.lib
    ident = ctypes.create_string_buffer(eng.COMM_ID_BYTES)
    assert lib.dffw_comm_unique_id(ident) == 0, lib.dffw_last_error()
    comm = ctypes.c_void_p()
    assert lib.dffw_comm_init_rank(0, 1, 0, ident.raw, ctypes.byref(comm)) == 0, lib.dffw_last_error()
    assert lib.dffw_comm_rank(comm) == 0 and lib.dffw_comm_size(comm) == 1
    send = torch.rand(4, 64, 96, device="cuda")
    recv = torch.zeros_like(send)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.dffw_allgather(comm, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()), send.numel(), ctypes.c_void_p(s)) == 0
    torch.cuda.synchronize()
    assert torch.equal(send, recv)
    lib.dffw_comm_destroy(comm)


_RANK_SCRIPT = r"""
import ctypes, os, sys, time, torch
sys.path.insert(0, sys.argv[1])
rank, world, idfile = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
from dffinthewild_amd import engine as eng
lib = eng.lib
torch.cuda.set_device(rank)
if rank == 0:
    ident = ctypes.create_string_buffer(eng.COMM_ID_BYTES)
    assert lib.dffw_comm_unique_id(ident) == 0, lib.dffw_last_error()
    with open(idfile + ".tmp", "wb") as f:
        f.write(ident.raw)
    os.replace(idfile + ".tmp", idfile)
    raw = ident.raw
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 120
        time.sleep(0.05)
    raw = open(idfile, "rb").read()
comm = ctypes.c_void_p()
assert lib.dffw_comm_init_rank(rank, world, rank, raw, ctypes.byref(comm)) == 0, lib.dffw_last_error()
b, H, W = 3, 32, 64
send = torch.full((b, H, W), float(rank + 1), device="cuda") + torch.arange(b, device="cuda").reshape(b, 1, 1) * 0.25
recv = torch.zeros((world * b, H, W), device="cuda")
s = torch.cuda.current_stream().cuda_stream
assert lib.dffw_allgather(comm, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()), send.numel(), ctypes.c_void_p(s)) == 0
torch.cuda.synchronize()
for r in range(world):
    want = torch.full((b, H, W), float(r + 1), device="cuda") + torch.arange(b, device="cuda").reshape(b, 1, 1) * 0.25
    assert torch.equal(recv[r * b:(r + 1) * b], want), (rank, r)
lib.dffw_comm_destroy(comm)
print("rank", rank, "ok")
"""


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one process per GPU)")
def test_allgather_two_ranks_one_process_per_gpu(lib_built, tmp_path):
    idfile = str(tmp_path / "rccl_id")
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", idfile], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
