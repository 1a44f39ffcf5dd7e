"""-m gpu: the RCCL code path of the sharded csrgemv 'T' (the ONE collective of the hot path: the reference's
mutex-guarded vector add, include/tasks/csrgemv_task.h:169-176) executed on the GPU -- in a world of ONE rank,
which is all a 1-GPU box can offer: init_process_group("nccl"), all_reduce, reduce_scatter + all_gather, and
bof_dist.flash_csrgemv_row_sharded with the reduce on the device.  The N > 1 behaviour is covered by the gloo
tests (tests/test_dist_gloo.py); this test makes sure the nccl backend itself loads, initialises and moves data
on this image (HSA_ENABLE_IPC_MODE_LEGACY=0 and all)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bofhip, bof_dist, orc
bofhip.require_device()
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{PORT}", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
t = torch.arange(1 << 20, dtype=torch.float32, device="cuda:0")
want = t.clone()
dist.all_reduce(t)
torch.cuda.synchronize()
assert torch.equal(t, want)
for algo in ("allreduce", "rs_ag"):
    for n in (1 << 20, 1000003):          # a length the world size divides, and one that needs the padded scratch
        y = torch.arange(n, dtype=torch.float32, device="cuda:0") % 977
        w = y.clone()
        bof_dist.allreduce_partial(y, algo=algo, force=True)
        torch.cuda.synchronize()
        assert torch.equal(y, w), (algo, n)
# the sharded csrgemv 'T' end to end with the reduce on the device (exact: integer generator data)
m, n = 4096, 2048
val, ja, ia = orc.sparse_create(m, n, 0.01)
d = os.environ["BOF_TEST_DIR"]
paths = {k: os.path.join(d, k) for k in ("val", "ja", "ia")}
val.tofile(paths["val"]); ja.tofile(paths["ja"]); ia.tofile(paths["ia"])
fds = {k: os.open(p, os.O_RDONLY) for k, p in paths.items()}
x = (np.arange(m) % 10).astype(np.float32)
y = np.zeros(n, np.float32)
bof_dist.flash_csrgemv_row_sharded("T", m, n, fds["val"], fds["ia"], fds["ja"], x, y, ia,
                                   bofhip.default_options(max_nnzs=5000, csrmm_rblk=1000), reduce_device="cuda:0")
ref = orc.flash_csrgemv("T", m, n, val, ia, ja, x, np.zeros(n, np.float32), 1000, 5000)
assert np.array_equal(y, ref)
for fd in fds.values():
    bofhip.lib().bof_file_forget(fd); os.close(fd)
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK backend=nccl")
"""


def test_rccl_world_of_one(dev, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BOF_TEST_DIR=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    code = f"ROOT = {ROOT!r}\nPORT = {port}\n" + CHILD
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
