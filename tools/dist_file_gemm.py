#!/usr/bin/env python3
"""Row-sharded multi-GPU flash::gemm on files (bof_dist.flash_gemm_row_sharded), one process per GPU:

    python -m torch.distributed.run --nproc-per-node N tools/dist_file_gemm.py A.bin B.bin C.bin m n k alpha beta [lda ldb ldc blk]

BOF_BENCH_ONE_GPU=1 puts every rank on cuda:0 with the gloo backend (the N > 1 path on a single-GPU
box; the only torch.distributed calls are barriers)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bofhip  # noqa: E402
import bof_dist  # noqa: E402


def main():
    pa, pb, pc = sys.argv[1:4]
    m, n, k = (int(v) for v in sys.argv[4:7])
    alpha, beta = float(sys.argv[7]), float(sys.argv[8])
    lda, ldb, ldc, blk = (int(v) for v in (sys.argv[9:13] + ["0"] * 4)[:4])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    one_gpu = os.environ.get("BOF_BENCH_ONE_GPU", "0") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    bofhip.require_device()
    flags = os.O_RDWR | (os.O_DIRECT if os.environ.get("BOF_ODIRECT", "0") == "1" else 0)
    fds = [os.open(p, flags) for p in (pa, pb, pc)]
    opts = bofhip.default_options(gemm_blk=blk or 4096, use_odirect=int(os.environ.get("BOF_ODIRECT", "0")),
                                  io_chunk_mib=int(os.environ.get("BOF_IO_CHUNK_MIB", "32")))
    if world > 1:
        dist.barrier()
    t0 = time.time()
    # every rank runs the level-3 file pipeline on its slab; BOF_B_ONCE=1 (default): B is read from storage
    # once per node (panel l by rank l % world, published in node-shared memory), 0: by every rank
    st = bof_dist.flash_gemm_row_sharded(m, n, k, alpha, beta, fds[0], fds[1], fds[2], lda, ldb, ldc, opts,
                                         b_once=os.environ.get("BOF_B_ONCE", "1") == "1")
    for fd in fds:
        os.fsync(fd)
        os.close(fd)
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    rank = dist.get_rank() if world > 1 else 0
    for turn in range(world):          # one rank at a time, so the launcher never interleaves two lines
        if turn == rank:
            print(json.dumps({"rank": rank, "world": world, "seconds": round(dt, 3), **st}), flush=True)
        if world > 1:
            dist.barrier()
            time.sleep(0.05)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
