#!/usr/bin/env python3
"""Point-sharded multi-GPU flash::kmeans distance matrix on files (bof_dist.flash_kmeans_point_sharded), one
process per GPU:

    python -m torch.distributed.run --nproc-per-node N tools/dist_file_kmeans.py centers.bin points.bin dist.bin ncenters npoints dim [blk]

BOF_BENCH_ONE_GPU=1 puts every rank on cuda:0 (debugging the N > 1 path on a single-GPU box)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bofhip  # noqa: E402
import bof_dist  # noqa: E402


def main():
    pc, pp, pd = sys.argv[1:4]
    ncenters, npoints, dim = (int(v) for v in sys.argv[4:7])
    blk = int(sys.argv[7]) if len(sys.argv) > 7 else 4096
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("BOF_BENCH_ONE_GPU", "0") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")      # only barriers: the path has no collective
    bofhip.require_device()
    centers = np.fromfile(pc, np.float32).reshape(ncenters, dim)
    points = np.memmap(pp, np.float32, "r", shape=(npoints, dim))
    cl = (centers.astype(np.float64) ** 2).sum(1).astype(np.float32)
    pl = (np.asarray(points, np.float64) ** 2).sum(1).astype(np.float32)
    fds = [os.open(p, os.O_RDWR) for p in (pc, pp, pd)]
    opts = bofhip.default_options(gemm_blk=blk, use_odirect=0)
    if world > 1:
        dist.barrier()
    t0 = time.time()
    st = bof_dist.flash_kmeans_point_sharded(ncenters, npoints, dim, fds[0], fds[1], fds[2], cl, pl, opts)
    for fd in fds:
        os.fsync(fd)
        os.close(fd)
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    rank = dist.get_rank() if world > 1 else 0
    for turn in range(world):
        if turn == rank:
            print(json.dumps({"rank": rank, "world": world, "seconds": round(dt, 3), **st}), flush=True)
        if world > 1:
            dist.barrier()
            time.sleep(0.05)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
