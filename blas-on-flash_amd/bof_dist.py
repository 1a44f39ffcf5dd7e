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
    GPU.  The k-panels of the tile grid are dealt round-robin to the ranks; the owner reads its
    panels from the file and broadcasts them (RCCL over xGMI, asynchronous, in k order) while the
    A / C slabs stream in; the tile DAG of each k range is queued behind "its" broadcast only, so
    MFMA work on panel l overlaps the arrival of panels l+1, ...; same tiles, same k-order as the
    single call, so the C file is bit-identical.  Pays when the storage, not PCIe, is the
    bottleneck (8 x 16 GiB of B reads at cfg4: 144 GiB through one device vs 32 GiB).

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
    # --- k-panels of B: the tile grid's own k blocks (tail-merge rule of src/blas/gemm.cpp:69-75),
    # dealt round-robin to the ranks.  Owner reads its panels from the file; every panel is then
    # broadcast (RCCL over xGMI, asynchronous, in k order) and consumed by the tile DAG of THAT k
    # range as soon as it has landed -- the accumulate chains run l = 0, 1, ... exactly as in the
    # single call, so the C file is bit-identical.
    nkb = max(1, k // tile) if (k % tile) < 128 and k >= tile else k // tile + (1 if k % tile else 0)
    nkb = max(nkb, 1)
    kb = [(l * tile, k if l == nkb - 1 else (l + 1) * tile) for l in range(nkb)]
    b_elems = (k - 1) * ldb + n
    b_dev = torch.empty(b_elems, dtype=torch.float32, device=dev)

    def region(l):
        k0, k1 = kb[l]
        return k0 * ldb, ((k1 - k0 - 1) * ldb + n)          # element offset, element count

    mine = [l for l in range(nkb) if l % world == rank]
    for l in mine:
        off, cnt = region(l)
        bofhip.file_to_device(bofhip.FPtr(fd_b, off * 4), cnt * 4, b_dev.data_ptr() + off * 4, o, st)
        rd += cnt * 4
    works = [None] * nkb
    if world > 1:
        for l in range(nkb):
            off, cnt = region(l)
            if one_gpu_debug:          # two ranks on one device: the collective goes through host memory
                h = b_dev[off:off + cnt].cpu() if l % world == rank else torch.empty(cnt, dtype=torch.float32)
                dist.broadcast(h, src=l % world, group=group)
                if l % world != rank:
                    b_dev[off:off + cnt].copy_(h)
            else:
                works[l] = dist.broadcast(b_dev[off:off + cnt], src=l % world, group=group, async_op=True)
    stats = {"bytes_read": 0, "bytes_written": 0, "rows": rows,
             "b_panel_rows": sum(kb[l][1] - kb[l][0] for l in mine)}
    if rows > 0:
        # A / C slabs stream in while the broadcasts are in flight
        a_dev = torch.empty(rows * lda, dtype=torch.float32, device=dev)
        a_bytes = ((rows - 1) * lda + k) * 4
        bofhip.file_to_device(bofhip.FPtr(fd_a, r0 * lda * 4), a_bytes, a_dev.data_ptr(), o, st)
        rd += a_bytes
        c_dev = torch.empty(rows * ldc, dtype=torch.float32, device=dev)
        c_bytes = ((rows - 1) * ldc + n) * 4
        if beta != 0.0 or ldc != n:       # padded rows: keep what lies between the row ends
            bofhip.file_to_device(bofhip.FPtr(fd_c, r0 * ldc * 4), c_bytes, c_dev.data_ptr(), o, st)
            rd += c_bytes
        for l in range(nkb):
            k0, k1 = kb[l]
            if works[l] is not None:
                works[l].wait()           # the CURRENT STREAM waits for panel l; the host does not
            bofhip.gemm_resident("R", "N", "N", rows, n, k1 - k0, alpha, beta if l == 0 else 1.0,
                                 a_dev.data_ptr() + k0 * 4, b_dev.data_ptr() + k0 * ldb * 4, c_dev.data_ptr(),
                                 lda, ldb, ldc, o, st)
        # the write-back is ordered behind the DAG through its stream argument
        bofhip.device_to_file(bofhip.FPtr(fd_c, r0 * ldc * 4), c_bytes, c_dev.data_ptr(), o, st)
        wr += c_bytes
    else:
        for w in works:
            if w is not None:
                w.wait()
    torch.cuda.current_stream(dev).synchronize()
    stats["bytes_read"], stats["bytes_written"] = rd, wr
    return stats


def flash_kmeans_point_sharded(ncenters, npoints, dim, fd_centers, fd_points, fd_dist, c_l2sq, p_l2sq, opts=None,
                               group=None):
    """Multi-GPU flash::kmeans in the reference driver's call shape (drivers/kmeans.cpp:37-39:
    'C','T','N', alpha = -2, beta = 0; dist is column-major ncenters x npoints, i.e. one contiguous
    run of ncenters distances per point).  The distance matrix shards by POINTS -- the dimension its
    panels run along -- so rank g owns the tile-aligned points [p0, p1): its slice of the points file,
    of p_l2sq and of the dist file; the centres and c_l2sq are read by every rank (small).  No
    collective, like the row-sharded gemm.  c_l2sq / p_l2sq: host float32 arrays (numpy).
    Returns {points, bytes_read, bytes_written} of this rank."""
    import numpy as np
    import torch.distributed as dist
    import bofhip
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    o = opts if opts is not None else bofhip.default_options()
    tile = int(o.gemm_blk)
    p0, p1 = row_shard(npoints, world, rank, tile)
    stats = {"points": p1 - p0, "bytes_read": 0, "bytes_written": 0}
    if p1 > p0:
        ones = np.ones(max(min(ncenters, tile + 127), min(p1 - p0, tile + 127)), np.float32)
        pl = np.ascontiguousarray(p_l2sq[p0:p1], np.float32)
        cl = np.ascontiguousarray(c_l2sq, np.float32)
        bofhip.flash_kmeans("C", "T", "N", ncenters, p1 - p0, dim, -2.0, 0.0, bofhip.FPtr(fd_centers, 0),
                            bofhip.FPtr(fd_points, p0 * dim * 4), bofhip.FPtr(fd_dist, p0 * ncenters * 4), dim, dim,
                            ncenters, cl.ctypes.data, pl.ctypes.data, ones.ctypes.data, o)
        st_ = bofhip.flash_last_stats()
        stats["bytes_read"], stats["bytes_written"] = st_["bytes_read"], st_["bytes_written"]
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
