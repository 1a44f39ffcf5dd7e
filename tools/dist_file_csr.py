#!/usr/bin/env python3
"""Row-sharded multi-GPU flash::csrmm / csrgemv on files (bof_dist), one process per GPU:

    torchrun --nproc-per-node N tools/dist_file_csr.py DIR m n k

DIR holds A.csr A.col A.off B.bin C.bin (csrmm, alpha=1 beta=0 'R') and x.bin (length max(m,n)); the
tool writes C.bin in place, yN.bin (rank 0 gathers the disjoint slices) and yT.bin.
BOF_BENCH_ONE_GPU=1: every rank on cuda:0, collectives through gloo."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bofhip  # noqa: E402
import bof_dist  # noqa: E402


def main():
    d = sys.argv[1]
    m, n, k = (int(v) for v in sys.argv[2:5])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    one_gpu = os.environ.get("BOF_BENCH_ONE_GPU", "0") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if one_gpu else "nccl",
                                **({} if one_gpu else {"device_id": torch.device("cuda", local)}))
    rank = dist.get_rank() if world > 1 else 0
    bofhip.require_device()
    fd = {x: os.open(os.path.join(d, x), os.O_RDWR) for x in ("A.csr", "A.col", "A.off", "B.bin", "C.bin")}
    ia = np.fromfile(os.path.join(d, "A.off"), np.int64)
    xfull = np.fromfile(os.path.join(d, "x.bin"), np.float32)
    opts = bofhip.default_options(max_nnzs=int(os.environ.get("BOF_MAX_NNZS", "10000000")),
                                  csrmm_rblk=int(os.environ.get("BOF_CSRMM_RBLK_SIZE", "131072")), use_odirect=0)
    r0, r1 = bof_dist.flash_csrmm_row_sharded(m, n, k, 1.0, 0.0, fd["A.csr"], fd["A.off"], fd["A.col"], "R",
                                              fd["B.bin"], fd["C.bin"], ia, opts)
    yn = np.zeros(m, np.float32)
    bof_dist.flash_csrgemv_row_sharded("N", m, n, fd["A.csr"], fd["A.off"], fd["A.col"], xfull[:n].copy(), yn, ia, opts)
    if world > 1:   # the slices are disjoint and the rest is zero: a sum is a gather
        t = torch.from_numpy(yn)
        dist.all_reduce(t) if one_gpu else None
        if not one_gpu:
            tg = t.cuda()
            dist.all_reduce(tg)
            yn = tg.cpu().numpy()
    yt = np.zeros(n, np.float32)
    bof_dist.flash_csrgemv_row_sharded("T", m, n, fd["A.csr"], fd["A.off"], fd["A.col"], xfull[:m].copy(), yt, ia, opts,
                                       reduce_device=None if one_gpu or world == 1 else torch.device("cuda", local))
    for f in fd.values():
        os.fsync(f)
        os.close(f)
    if world > 1:
        dist.barrier()
    if rank == 0:
        yn.tofile(os.path.join(d, "yN.bin"))
        yt.tofile(os.path.join(d, "yT.bin"))
    for turn in range(world):
        if turn == rank:
            print(json.dumps({"rank": rank, "world": world, "rows": [int(r0), int(r1)]}), flush=True)
        if world > 1:
            dist.barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
