"""Multi-GPU sharding of the hot path: one process per GPU (torch.distributed;
backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The path shards by OUTPUT ROW BLOCK: GEMM, CSRMM and CSRGEMV 'N'
need no exchange at all -- every rank owns a disjoint slab of C / y and calls the
single-GPU entry points on its slab.  The only collective is the partial-sum
reduce of CSRGEMV 'T' (the reference's mutex-guarded vector add,
include/tasks/csrgemv_task.h:169-176): each rank produces a full-length partial
y_g = A_g^T x_g and one all-reduce(sum) combines them.
"""
import math

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


def allreduce_partial(y, group=None, algo="allreduce", force=False):
    """Sum the per-rank partial vectors of CSRGEMV 'T' in place (RCCL over xGMI for device tensors).
    fp32; exact for the integer-valued generator data.

    algo "allreduce": one all-reduce (RCCL picks ring / tree).  algo "rs_ag": reduce-scatter then
    all-gather -- on the fully connected xGMI mesh (7 point-to-point links per GPU) every rank sends
    its 7 foreign chunks of S/8 over 7 different links at once, so the step is bound by S/8 per
    link instead of a ring's 2 (N-1)/N S over one.  The vector length is padded
    to a multiple of the world size in a scratch tensor when needed.
    force=True issues the collectives even in a world of one rank (the GPU smoke test of the RCCL path)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
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


_share_seq = 0


def flash_gemm_row_sharded(m, n, k, alpha, beta, fd_a, fd_b, fd_c, lda=0, ldb=0, ldc=0, opts=None, group=None,
                           b_once=True, local_slabs=False):
    """Multi-GPU flash::gemm('R','N','N') on FILE-resident matrices (BASELINE configs[3]), one
    process per GPU: rank g owns the C rows [r0, r1) (tile-aligned, `row_shard`) and calls the
    single-GPU file pipeline on its slab -- bof_flash_gemm with the A and C pointers advanced to row
    r0.  In level 3 that is the row-panel pipeline: the rank's A panels stream through a ring, C panels
    are written back while later ones compute; reads, PCIe copies, MFMA work and write-back all overlap
    inside the library (flash_gemm_panels.cpp), nothing is staged in Python, and there is no data-path
    collective (BASELINE configs[3]).

    b_once=True (default for world > 1): B is read from storage ONCE PER NODE instead of
    once per rank -- inside the same pipeline.  The ranks that own rows pass share_world / share_rank /
    share_name to bof_flash_gemm: panel l of B is read from the file by rank l % share_world, which
    publishes it chunk by chunk in a node-shared staging ring; the others take it from there (bof_options
    in include/bof_hip.h).  At cfg4 that is 16 + 16 GiB of reads per node instead of 16 + 8 x 16.  The
    only torch.distributed call is the barrier behind which the staging ring is removed.
    b_once=False: every rank reads B itself.
    local_slabs=True: fd_a / fd_c are THIS RANK'S OWN files holding only its rows (A rows [r0, r1) at offset 0, its
    C slab likewise) instead of the whole matrices -- how a node with one scratch volume per GPU spreads the file
    set (bench.py: $BOF_BENCH_DIRS); fd_b is then the rank's replica of B.

    Returns {rows, bytes_read, bytes_written, bytes_peer, seconds} of this rank."""
    global _share_seq
    import os
    import torch.distributed as dist
    import bofhip
    lda, ldb, ldc = lda or k, ldb or n, ldc or n
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if world > 1 else 0
    o = opts if opts is not None else bofhip.default_options()
    tile = int(o.gemm_blk)
    r0, r1 = row_shard(m, world, rank, tile)
    rows = r1 - r0
    stats = {"rows": rows, "bytes_read": 0, "bytes_written": 0, "bytes_peer": 0, "seconds": 0.0}
    # the ranks that own rows, in rank order (the same list on every rank)
    owners = [g for g in range(world) if row_shard(m, world, g, tile)[1] > row_shard(m, world, g, tile)[0]]
    name, mates, leader = None, [], None
    if world > 1 and b_once and len(owners) > 1:
        _share_seq += 1                      # every rank makes the same sequence of calls
        # The staging ring is POSIX shared memory: a share group = the row-owning ranks of ONE host.  Every rank
        # tells the others where it runs (host name + boot id: two containers of one machine with separate /dev/shm
        # differ in neither, which is why the host's first owner also PROBES the segment below).
        hosts = _hosts_of(group, world)
        mates = [g for g in owners if hosts[g] == hosts[rank]] if rank in owners else []
        # ... and the host's first owner decides whether its /dev/shm can hold the ring (64 chunk slots per shared
        # operand, ~2 GiB at the default 32 MiB chunk; a store into a tmpfs page that cannot be allocated is a
        # SIGBUS, so with room to spare only) and removes leftovers of a crashed earlier run.  The smallest group
        # rank of the share group is part of the name: disjoint groups never meet in one segment.
        verdict = None
        if len(mates) > 1:
            leader = mates[0]
            name = f"/bof_{os.environ.get('MASTER_PORT', '0')}_{os.getuid()}_{leader}_{_share_seq}"
            if rank == leader:
                bofhip.lib().bof_share_cleanup(name.encode())
                try:
                    sv = os.statvfs("/dev/shm")
                    ring = 2 * 64 * ((max(int(o.io_chunk_mib), 1) << 20) + 8192)
                    verdict = bool(sv.f_bavail * sv.f_frsize > ring * 1.25 + (64 << 20))
                except OSError:
                    verdict = False
        # Can THIS rank's call take the row-panel path?  Layout reasons (C rows with gaps, ...) are the same on every
        # rank, the HBM budget is not (it is 0.8 of what is free on the rank's own GPU).  The staging ring only works if
        # every member of the share group is on the panel path, so the ranks agree BEFORE the call: one member that
        # cannot -> the whole group reads B itself (and that member's call goes to the tile cache).  The library's own
        # check (BOF_EINVAL + the group's failure word) stays as the safety net behind this.
        can, pre_err = None, None
        if rows > 0:
            # (an exception here would leave the other ranks blocked in the gather below: it travels WITH the verdict
            #  and is raised on every rank afterwards, like the post-call errors)
            try:
                can = int(o.gemm_path) != 1
                if can:
                    budget = int(o.hbm_budget)
                    if budget <= 0:
                        import ctypes
                        fr, tot = ctypes.c_size_t(), ctypes.c_size_t()
                        bofhip.check(bofhip.lib().bof_mem_info(ctypes.byref(fr), ctypes.byref(tot)), "bof_mem_info")
                        budget = int(fr.value * 0.8)
                    pl = bofhip.flash_gemm_panel_plan("R", "N", "N", rows, n, k, tile, budget, lda, ldb, ldc, 0)
                    # beta != 0: the ramp group's chains also need their raw accumulator panels (bof_panel_plan.acc_bytes).
                    # The library sets them aside BEFORE spare budget deepens the C ring (plan_panels(..., with_acc),
                    # plan.cpp), while the plan asked for here has already spent the spare budget on the ring and reports
                    # need_bytes behind that -- so the test is made against the MINIMAL rings (2 * group + 1 C slots), as
                    # PanelRun::plan does; need_bytes + acc_bytes could refuse a call the library would take.
                    min_c = min(pl["n_panels"][2], 2 * pl["first_group"] + 1)
                    need_min = (pl["n_slots"][0] * pl["slot_bytes"][0] + pl["n_slots"][1] * pl["slot_bytes"][1]
                                + min_c * pl["slot_bytes"][2])
                    can = bool(pl["eligible"]) and (beta == 0 or int(o.gemm_chain) == 1 or need_min + pl["acc_bytes"] <= budget)
            except Exception as e:      # noqa: BLE001 -- whatever it is, every rank must hear of it
                can, pre_err = False, f"{type(e).__name__}: {e}"
        verdicts = [None] * world
        dist.all_gather_object(verdicts, (verdict, can, pre_err), group=group)
        bad_pre = [(g, v[2]) for g, v in enumerate(verdicts) if v is not None and v[2]]
        if bad_pre:
            raise bofhip.BofError(f"flash_gemm_row_sharded: rank {bad_pre[0][0]} failed its eligibility check: {bad_pre[0][1]}")
        if name is not None and not verdicts[leader][0]:
            name = None
            stats["b_once"] = "off: /dev/shm of this host cannot hold the staging ring"
        elif name is not None and not all(verdicts[g][1] for g in mates):
            name = None
            stats["b_once"] = "off: a rank of this host cannot take the row-panel path (layout or its HBM budget)"
        elif name is None and rank in owners:
            stats["b_once"] = "off: the only row-owning rank of its host"
    err = None
    if rows > 0:
        import ctypes
        call = bofhip.Options()
        ctypes.memmove(ctypes.byref(call), ctypes.byref(o), ctypes.sizeof(o))
        if name is not None:
            call.share_world, call.share_rank, call.share_name = len(mates), mates.index(rank), name.encode()
        try:
            off_a, off_c = (0, 0) if local_slabs else (r0 * lda * 4, r0 * ldc * 4)
            bofhip.flash_gemm("R", "N", "N", rows, n, k, alpha, beta, bofhip.FPtr(fd_a, off_a),
                              bofhip.FPtr(fd_b, 0), bofhip.FPtr(fd_c, off_c), lda, ldb, ldc, call)
            st_ = bofhip.flash_last_stats()
            for q in ("bytes_read", "bytes_written", "bytes_peer", "seconds", "kernel_launches", "kernel_seconds", "tasks",
                      "bytes_h2d", "bytes_d2h"):
                stats[q] = st_[q]
        except bofhip.BofError as e:
            err = str(e)
    if world > 1 and b_once and len(owners) > 1:
        # everybody is out of the call -- whether it worked or not: a rank that failed must not leave the others
        # in a barrier, so the ranks exchange their status (this IS the barrier), the leaders remove the
        # segments, and only then does a failure become an exception -- on every rank
        errs = [None] * world
        dist.all_gather_object(errs, err, group=group)
        if name is not None and rank == leader:
            bofhip.lib().bof_share_cleanup(name.encode())
        bad = [(g, e) for g, e in enumerate(errs) if e]
        if bad:
            raise bofhip.BofError(err if err else f"flash_gemm_row_sharded: rank {bad[0][0]} failed: {bad[0][1]}")
    elif err:
        raise bofhip.BofError(err)
    return stats


_host_id = None
_hosts_cache = {}


def _host_identity():
    global _host_id
    if _host_id is None:
        import socket
        boot = ""
        try:
            boot = open("/proc/sys/kernel/random/boot_id").read().strip()
        except OSError:
            pass
        _host_id = socket.gethostname() + "/" + boot
    return _host_id


def _hosts_of(group, world):
    """Where every rank of the group runs; gathered once per (group, world), not per call."""
    import torch.distributed as dist
    key = (id(group) if group is not None else None, world)
    if key not in _hosts_cache:
        hosts = [None] * world
        dist.all_gather_object(hosts, _host_identity(), group=group)
        _hosts_cache[key] = hosts
    return _hosts_cache[key]


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
    # the cuts between ranks fall on PAGES of the C file (r0 * k * 4 a multiple of 4096): each rank's call picks O_DIRECT
    # or the page cache for C by the alignment of ITS row blocks, and a direct write must never meet a neighbour's
    # buffered write in one page (csrc/flash_csr.cpp, c_blocks_aligned)
    r0, r1 = csr_row_shard(ia_host, world, rank, max(128, 1024 // math.gcd(int(k), 1024)))
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
