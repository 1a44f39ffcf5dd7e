"""-m gpu: the in-HBM input generators against the oracle (itself pinned to the
reference tools' known-answer hashes)."""
import hashlib

import numpy as np
import pytest
import torch

import bofhip
import orc
from gpu_util import ptr, stream

pytestmark = pytest.mark.gpu


def h16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


@pytest.mark.parametrize("nrows,ncols,sp,kat", [
    (64, 1000, 0.01, ("d8503dd0b005b159", "5c25d609762b16cd", "6cfa6d1ddb47f16a")),
    (1000, 100000, 0.0001, ("4ab90aabd38c5029", "df0b4409a7318e45", "f0510c987daf1cfe")),
    (4096, 2048, 0.01, ("0f7f6e93eeac0fdb", "19d3d33d2e9e0cfc", "2b8ca0ee3390fec1")),
])
def test_sparse_generator_kat(dev, nrows, ncols, sp, kat):
    npr = orc.lib().orc_sparse_nnz_per_row(ncols, sp)
    csr = torch.empty(nrows * npr, dtype=torch.float32, device=dev)
    col = torch.empty(nrows * npr, dtype=torch.int64, device=dev)
    off = torch.empty(nrows + 1, dtype=torch.int64, device=dev)
    bofhip.gen_sparse_rows(0, nrows, ncols, npr, ptr(csr), ptr(col), ptr(off), stream())
    torch.cuda.synchronize()
    assert (h16(csr.cpu().numpy()), h16(col.cpu().numpy()), h16(off.cpu().numpy())) == kat
    v, c, o = orc.sparse_create(nrows, ncols, sp)
    assert np.array_equal(col.cpu().numpy(), c)


def test_sparse_generator_chunked(dev):
    """Row-range generation (used for the 10M-row cfg3 matrix) equals one-shot."""
    nrows, ncols, npr = 5000, 1000000, 100
    v, c, o = orc.sparse_create(nrows, ncols, 0.0001)
    csr = torch.empty(nrows * npr, dtype=torch.float32, device=dev)
    col = torch.empty(nrows * npr, dtype=torch.int64, device=dev)
    off = torch.empty(nrows + 1, dtype=torch.int64, device=dev)
    for r0 in (0, 1234, 4000):
        r1 = {0: 1234, 1234: 4000, 4000: 5000}[r0]
        bofhip.gen_sparse_rows(r0, r1 - r0, ncols, npr, ptr(csr) + 4 * r0 * npr,
                               ptr(col) + 8 * r0 * npr, ptr(off) + 8 * r0, stream())
    torch.cuda.synchronize()
    assert np.array_equal(col.cpu().numpy(), c)
    assert np.array_equal(csr.cpu().numpy(), v)
    assert np.array_equal(off.cpu().numpy(), o)
    assert c[:8].tolist() == [2180, 20218, 49198, 51205, 60867, 67543, 68547, 69224]  # App. A-3 cfg3 row 0


def test_dense_generator(dev):
    d = torch.empty(37 * 53, dtype=torch.float32, device=dev)
    bofhip.gen_dense(ptr(d), 0, d.numel(), "s", 0, stream())
    torch.cuda.synchronize()
    assert h16(d.cpu().numpy()) == "d08eb5a3728513a6"   # dense_create d.bin 37 53 s (App. A-3)
    bofhip.gen_dense(ptr(d), 1000, d.numel(), "s", 0, stream())
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), ((1000 + np.arange(d.numel())) % 10).astype(np.float32))
    bofhip.gen_dense(ptr(d), 0, d.numel(), "u", 42, stream())
    torch.cuda.synchronize()
    u = d.cpu().numpy()
    assert u.min() >= -1.0 and u.max() < 1.0 and abs(u.mean()) < 0.1 and len(np.unique(u)) > 1900
