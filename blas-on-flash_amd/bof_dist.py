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


def allreduce_partial(y, group=None):
    """Sum the per-rank partial vectors of CSRGEMV 'T' in place (RCCL all-reduce over
    xGMI for device tensors).  fp32; exact for the integer-valued generator data."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
    return y
