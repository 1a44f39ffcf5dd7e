"""Multi-GPU sharding of the hot path: one process per GPU (torch.distributed;
backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The path shards by OUTPUT ROW BLOCK: GEMM, CSRMM and CSRGEMV 'N'
need no exchange at all -- every rank owns a disjoint slab of C / y and calls the
single-GPU entry points on its slab.  The only collective is the partial-sum
reduce of CSRGEMV 'T' (the reference's mutex-guarded vector add,
include/tasks/csrgemv_task.h:169-176): each rank produces a full-length partial
y_g = A_g^T x_g and one all-reduce(sum) combines them.
"""
import numpy as np


def row_shard(m, world, rank, align=1):
    """Contiguous row range [r0, r1) of rank `rank`; boundaries are multiples of
    `align` (use the tile edge so shard boundaries coincide with tile boundaries)."""
    units = (m + align - 1) // align
    base, extra = divmod(units, world)
    u0 = rank * base + min(rank, extra)
    u1 = u0 + base + (1 if rank < extra else 0)
    return min(u0 * align, m), min(u1 * align, m)


def csr_row_shard(ia, world, rank, align=1):
    """Row range balanced by non-zeros: boundaries at the rows where the running nnz
    crosses g/world of the total (rounded to `align` rows)."""
    ia = np.asarray(ia)
    m = ia.size - 1
    total = int(ia[m] - ia[0])

    def cut(g):
        if g <= 0:
            return 0
        if g >= world:
            return m
        r = int(np.searchsorted(ia, ia[0] + total * g // world, side="left"))
        r = (r + align // 2) // align * align
        return max(0, min(m, r))

    r0, r1 = cut(rank), cut(rank + 1)
    return r0, max(r0, r1)


def gemm_shard_args(m, n, k, lda, ldc, world, rank, tile):
    """Arguments of the per-rank flash::gemm call for row-major 'N','N':
    (m_local, element offset into A, element offset into C)."""
    r0, r1 = row_shard(m, world, rank, tile)
    return r1 - r0, r0 * (lda or k), r0 * (ldc or n)


def allreduce_partial(y, group=None, algo="allreduce"):
    """Sum the per-rank partial vectors of CSRGEMV 'T' in place (RCCL over xGMI for device tensors).
    fp32; exact for the integer-valued generator data.

    algo "allreduce": one all-reduce (RCCL picks ring / tree).  algo "rs_ag": reduce-scatter then
    all-gather -- on the fully connected xGMI mesh (7 point-to-point links per GPU) every rank sends
    its 7 foreign chunks of S/8 over 7 different links at once, so the step is bound by S/8 per
    link instead of a ring's 2 (N-1)/N S over one.  The vector length is padded
    to a multiple of the world size in a scratch tensor when needed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return y
    world = dist.get_world_size(group)
    if algo != "rs_ag":
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y
    n = y.numel()
    per = (n + world - 1) // world
    src = y
    if per * world != n:
        src = torch.zeros(per * world, dtype=y.dtype, device=y.device)
        src[:n] = y
    mine = torch.empty(per, dtype=y.dtype, device=y.device)
    dist.reduce_scatter_tensor(mine, src, op=dist.ReduceOp.SUM, group=group)
    dist.all_gather_into_tensor(src, mine, group=group)
    if src is not y:
        y.copy_(src[:n])
    return y


def flash_gemm_row_sharded(m, n, k, alpha, beta, fd_a, fd_b, fd_c, lda=0, ldb=0, ldc=0, opts=None, group=None,
                           device=None, one_gpu_debug=False, b_once_per_node=False):
    """Multi-GPU flash::gemm('R','N','N') on FILE-resident matrices (BASELINE configs[3]):
    rank g owns the C rows [r0, r1) (tile-aligned, `row_shard`).

    Default (no collective, as BASELINE configs[3] says): every rank simply calls the single-GPU
    file pipeline on its slab -- bof_flash_gemm with the A and C pointers advanced to row r0.  In
    level 3 that is the row-panel pipeline: the rank's A panels stream through a ring, B is read by
    the rank itself (large sequential requests; from the page cache once another rank has touched
    it), C panels are written back while later ones compute -- reads, PCIe copies, MFMA work and
    write-back all overlap inside the library (flash_gemm_panels.cpp), nothing is staged in Python.

    b_once_per_node=True: B is read from storage ONCE per node instead of once per
    GPU: rank g reads the k-row panel B[k0:k1, :] and one all-gather (RCCL over xGMI) assembles the
    full matrix on every GPU; A / C slabs are made resident and the level-2 tile DAG runs over them
    (same tiles, same k-order, so the C file is bit-identical).  Pays when the storage, not PCIe, is
    the bottleneck (8 x 16 GiB of B reads at cfg4).

    Returns {bytes_read, bytes_written, rows, b_panel_rows} of this rank.  `one_gpu_debug` runs the
    collective through host memory (gloo) so that two ranks can share one device."""
    import torch
    import torch.distributed as dist
    import bofhip
    lda, ldb, ldc = lda or k, ldb or n, ldc or n
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    o = opts if opts is not None else bofhip.default_options()
    tile = int(o.gemm_blk)
    r0, r1 = row_shard(m, world, rank, tile)
    rows = r1 - r0
    if not b_once_per_node:
        stats = {"bytes_read": 0, "bytes_written": 0, "rows": rows, "b_panel_rows": k if rows > 0 else 0}
        if rows > 0:
            bofhip.flash_gemm("R", "N", "N", rows, n, k, alpha, beta, bofhip.FPtr(fd_a, r0 * lda * 4),
                              bofhip.FPtr(fd_b, 0), bofhip.FPtr(fd_c, r0 * ldc * 4), lda, ldb, ldc, o)
            st_ = bofhip.flash_last_stats()
            stats["bytes_read"], stats["bytes_written"] = st_["bytes_read"], st_["bytes_written"]
            stats["seconds"] = st_["seconds"]
        return stats
    st = torch.cuda.current_stream(dev).cuda_stream
    rd = wr = 0
    # --- B: this rank's k-row panel from the file, then all-gather ---------------------------
    per = (k + world - 1) // world
    k0, k1 = min(k, rank * per), min(k, (rank + 1) * per)
    # torch.empty: no fill kernel is queued on torch's stream that could land on top of the data
    # the library's private copy stream writes; only what the file does not cover is zeroed, and
    # the transfer itself is ordered behind `st` (bof_file_to_device's stream argument)
    panel = torch.empty(per * ldb, dtype=torch.float32, device=dev)
    nbytes = ((k1 - k0 - 1) * ldb + n) * 4 if k1 > k0 else 0   # the last row may end before a full ldb
    if nbytes < panel.numel() * 4:
        panel[nbytes // 4:].zero_()
    if k1 > k0:
        bofhip.file_to_device(bofhip.FPtr(fd_b, k0 * ldb * 4), nbytes, panel.data_ptr(), o, st)
        rd += nbytes
    if world > 1:
        full = torch.empty(world * per * ldb, dtype=torch.float32, device=dev)
        if one_gpu_debug:
            hp, hf = panel.cpu(), torch.empty(world * per * ldb, dtype=torch.float32)
            dist.all_gather_into_tensor(hf, hp, group=group)
            full.copy_(hf)
        else:
            dist.all_gather_into_tensor(full, panel, group=group)
        b_dev = full
    else:
        b_dev = panel
    stats = {"bytes_read": 0, "bytes_written": 0, "rows": rows, "b_panel_rows": k1 - k0}
    if rows > 0:
        a_dev = torch.empty(rows * lda, dtype=torch.float32, device=dev)
        a_bytes = ((rows - 1) * lda + k) * 4
        bofhip.file_to_device(bofhip.FPtr(fd_a, r0 * lda * 4), a_bytes, a_dev.data_ptr(), o, st)
        rd += a_bytes
        c_dev = torch.empty(rows * ldc, dtype=torch.float32, device=dev)
        c_bytes = ((rows - 1) * ldc + n) * 4
        if beta != 0.0 or ldc != n:       # padded rows: keep what lies between the row ends
            bofhip.file_to_device(bofhip.FPtr(fd_c, r0 * ldc * 4), c_bytes, c_dev.data_ptr(), o, st)
            rd += c_bytes
        # (the transfers above are blocking: the slabs are complete before the DAG is queued;
        #  the write-back below is ordered behind the DAG through its stream argument)
        bofhip.gemm_resident("R", "N", "N", rows, n, k, alpha, beta, a_dev.data_ptr(), b_dev.data_ptr(),
                             c_dev.data_ptr(), lda, ldb, ldc, o, st)
        bofhip.device_to_file(bofhip.FPtr(fd_c, r0 * ldc * 4), c_bytes, c_dev.data_ptr(), o, st)
        wr += c_bytes
    stats["bytes_read"], stats["bytes_written"] = rd, wr
    return stats


def _world_rank(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def flash_csrmm_row_sharded(m, n, k, alpha, beta, fd_a, fd_ia, fd_ja, ord_b, fd_b, fd_c, ia_host, opts=None,
                            group=None):
    """Multi-GPU flash::csrmm('N') on files: rank g owns the row range [r0, r1) balanced by
    non-zeros (`csr_row_shard`, 128-row aligned).  The shard is the offsets pointer advanced by
    r0 (offsets stay absolute, src/blas/csrmm.cpp:97-98); B is read by every rank (512 MB at
    BASELINE configs[2]); C rows are disjoint -- no collective.  Row-major C only ('R'): a
    column-major C slab is not contiguous in the file."""
    import bofhip
    if ord_b != "R":
        raise ValueError("flash_csrmm_row_sharded: ord_b must be 'R'")
    world, rank = _world_rank(group)
    r0, r1 = csr_row_shard(ia_host, world, rank, 128)
    if r1 > r0:
        bofhip.flash_csrmm("N", r1 - r0, n, k, alpha, beta, bofhip.FPtr(fd_a, 0), bofhip.FPtr(fd_ia, r0 * 8),
                           bofhip.FPtr(fd_ja, 0), "R", bofhip.FPtr(fd_b, 0), bofhip.FPtr(fd_c, r0 * k * 4), opts)
    return r0, r1


def flash_csrgemv_row_sharded(trans, m, n, fd_a, fd_ia, fd_ja, x, y, ia_host, opts=None, group=None,
                              reduce_device=None):
    """Multi-GPU flash::csrgemv on files with host vectors x, y (numpy fp32).  'N': rank g fills
    y[r0:r1] (disjoint; the caller gathers if it wants y everywhere).  'T': every rank computes the
    full-length partial of its rows and ONE all-reduce(sum) -- the only collective of the whole
    path -- leaves y = A^T x on every rank (`reduce_device`: torch device the reduce runs on;
    None = host/gloo)."""
    import numpy as np
    import torch
    import bofhip
    world, rank = _world_rank(group)
    r0, r1 = csr_row_shard(ia_host, world, rank, 128)
    rows = r1 - r0
    if trans == "N":
        if rows > 0:
            part = np.zeros(rows, np.float32)
            bofhip.flash_csrgemv("N", rows, n, bofhip.FPtr(fd_a, 0), bofhip.FPtr(fd_ia, r0 * 8),
                                 bofhip.FPtr(fd_ja, 0), x.ctypes.data, part.ctypes.data, opts)
            y[r0:r1] = part
        return r0, r1
    part = np.zeros(n, np.float32)
    if rows > 0:
        xs = np.ascontiguousarray(x[r0:r1])
        bofhip.flash_csrgemv("T", rows, n, bofhip.FPtr(fd_a, 0), bofhip.FPtr(fd_ia, r0 * 8),
                             bofhip.FPtr(fd_ja, 0), xs.ctypes.data, part.ctypes.data, opts)
    t = torch.from_numpy(part)
    if reduce_device is not None:
        t = t.to(reduce_device)
    allreduce_partial(t, group)
    y[:] = t.cpu().numpy()
    return r0, r1
