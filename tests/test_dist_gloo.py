"""world_size-2 gloo tests (CPU) of the multi-GPU sharding plumbing: row-block
shards are disjoint and cover the output, and the CSRGEMV 'T' partial-sum
all-reduce reproduces the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bof_dist  # noqa: E402
import orc  # noqa: E402


def test_row_shard_covers_and_aligns():
    for m, world, align in [(65536, 8, 4096), (32768, 8, 4096), (1000, 3, 128), (5, 8, 1), (640, 2, 256)]:
        spans = [bof_dist.row_shard(m, world, r, align) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == m
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            assert a1 == b0 and a0 <= a1
        assert all(s[0] % align == 0 for s in spans)
    assert bof_dist.row_shard(65536, 8, 3, 4096) == (3 * 8192, 4 * 8192)      # cfg4: 8192 rows each
    assert bof_dist.gemm_shard_args(65536, 65536, 65536, 0, 0, 8, 2, 4096) == (8192, 2 * 8192 * 65536, 2 * 8192 * 65536)


def test_csr_row_shard_balances_nnz():
    rng = np.random.default_rng(0)
    ia = np.concatenate([[0], np.cumsum(rng.integers(0, 50, 10000))]).astype(np.int64)
    spans = [bof_dist.csr_row_shard(ia, 4, r) for r in range(4)]
    assert spans[0][0] == 0 and spans[-1][1] == 10000
    nnz = [ia[b] - ia[a] for a, b in spans]
    assert max(nnz) - min(nnz) < 100
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, n = 4096, 2048
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    x = (np.arange(m) % 10).astype(np.float32)
    r0, r1 = bof_dist.csr_row_shard(ia, world, rank, align=128)
    # this rank's partial y_g = A_g^T x_g (on the GPU box: bof_flash_csrgemv 'T' on the slab)
    part = np.zeros(n, np.float32)
    # shard = offset `ia` pointer; val/ja stay whole-file (absolute offsets, as flash_ptr + r0)
    orc.flash_csrgemv("T", r1 - r0, n, val, ia[r0:r1 + 1], ja, x[r0:r1], part, 1000, 5000)
    y = torch.from_numpy(part)
    # the reduce-scatter + all-gather form must give the same vector (gloo builds without
    # reduce_scatter are skipped on that leg); n + 1 elements exercise the padding path
    y2 = torch.cat([y.clone(), torch.full((1,), float(rank + 1))])
    try:
        bof_dist.allreduce_partial(y2, algo="rs_ag")
        rs_ok = True
    except (RuntimeError, NotImplementedError):
        rs_ok = False
    bof_dist.allreduce_partial(y)
    if rs_ok:
        assert torch.equal(y2[:-1], y) and float(y2[-1]) == world * (world + 1) / 2
    np.save(os.path.join(out_dir, f"rs_{rank}.npy"), np.array([1 if rs_ok else 0]))
    # 'N' needs no collective: disjoint slices
    xn = (np.arange(n) % 10).astype(np.float32)
    yn = np.zeros(r1 - r0, np.float32)
    orc.flash_csrgemv("N", r1 - r0, n, val, ia[r0:r1 + 1], ja, xn, yn, 1000, 5000)
    np.save(os.path.join(out_dir, f"T_{rank}.npy"), y.numpy())
    np.save(os.path.join(out_dir, f"N_{rank}.npy"), np.concatenate([[r0, r1], yn]))
    dist.barrier()
    dist.destroy_process_group()


def test_csrgemv_two_rank_reduce(tmp_path, golden):
    import hashlib
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = {t.split()[1]: t.split()[2] for t in golden["meta"] if t.startswith("exact")}
    for r in range(world):
        assert int(np.load(tmp_path / f"rs_{r}.npy")[0]) == 1     # reduce-scatter + all-gather leg ran
        yT = np.load(tmp_path / f"T_{r}.npy")
        assert hashlib.sha256(yT.tobytes()).hexdigest() == want["gen_csrgemv_T"]
    parts = [np.load(tmp_path / f"N_{r}.npy") for r in range(world)]
    parts.sort(key=lambda p: p[0])
    assert parts[0][1] == parts[1][0]
    yN = np.concatenate([p[2:] for p in parts]).astype(np.float32)
    assert hashlib.sha256(yN.tobytes()).hexdigest() == want["gen_csrgemv_N"]


# ---- node-shared staging ring of the one-process-per-GPU GEMM (share_world > 1): host code only ---------
def _ring_worker(rank, world, name, n_chunks, chunk_bytes, n_slots, timeout_s, skip, q):
    import bofhip
    if rank in skip:                       # a rank that never shows up
        q.put((rank, None))
        return
    q.put((rank, int(bofhip.lib().bof_share_selftest(name.encode(), rank, world, n_chunks, chunk_bytes, n_slots,
                                                     timeout_s))))


def _run_ring(world, n_chunks, chunk_bytes, n_slots, timeout_s, skip=()):
    import bofhip
    name = f"/bof_test_{os.getpid()}_{world}_{n_slots}_{len(skip)}"
    bofhip.lib().bof_share_cleanup(name.encode())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ring_worker, args=(r, world, name, n_chunks, chunk_bytes, n_slots, timeout_s, skip, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    bofhip.lib().bof_share_cleanup(name.encode())
    assert not [f for f in os.listdir("/dev/shm") if f.startswith(name[1:])]
    return res


@pytest.mark.parametrize("world,n_slots", [(2, 1), (2, 8), (3, 4), (4, 64)])
def test_share_ring_delivers_every_chunk_once_to_every_peer(world, n_slots):
    """What bof_dist.flash_gemm_row_sharded relies on for "B once per node": chunk c is produced by rank
    c % world and taken by all others through a ring of n_slots reused slots (a slot is refilled only after
    every peer has taken its chunk); content checked chunk by chunk; no GPU involved."""
    n_chunks = 150
    res = _run_ring(world, n_chunks, 256 << 10, n_slots, 30.0)
    for r in range(world):
        own = len(range(r, n_chunks, world))
        assert res[r] == n_chunks - own, res


def test_share_ring_missing_peer_times_out():
    """A rank that never delivers: the others give up after the timeout with -ETIMEDOUT (or -EIO once a
    peer has marked the ring failed) instead of waiting forever."""
    res = _run_ring(3, 20, 64 << 10, 4, 1.5, skip=(1,))
    assert res[1] is None
    assert res[0] in (-110, -5) and res[2] in (-110, -5), res
