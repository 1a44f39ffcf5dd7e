#!/usr/bin/env python3
"""End-to-end FILE-resident flash::gemm (level 3) at a chosen size: the same measurement and the same
whole-file verification as bench.py's `e2e.gemm` block (bench.e2e_gemm), with the knobs exposed:
size, tile, descriptor mode, I/O path (tile cache / row panels), HBM budget."""
import argparse
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
import torch  # noqa: E402
import bofhip  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default=os.environ.get("TMPDIR", "/tmp"))
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--blk", type=int, default=4096)
    ap.add_argument("--direct", type=int, default=-1, help="1 O_DIRECT only, 0 buffered only, -1 both")
    ap.add_argument("--io-threads", type=int, default=8)
    ap.add_argument("--pinned", type=int, default=8)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--path", type=int, default=0, help="0 choose, 1 tile cache, 2 row panels")
    ap.add_argument("--chunk-mib", type=int, default=0)
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--budget-gib", type=float, default=0.0,
                    help="HBM budget (0 = library default: most of the free HBM); 8 = the reference's PROGRAM_BUDGET")
    ap.add_argument("--rank-calls", type=int, default=0,
                    help="run the W per-rank calls of a W-GPU row-sharded run one after the other (composition check)")
    ap.add_argument("--devices", default="", help="in-process device list, e.g. 0,0 (one GPU playing two devices)")
    ap.add_argument("--pre", default="", help="diagnostic: alloc,resident,dgemm,release stages run first")
    ap.add_argument("--cpu-warm", type=int, default=0,
                    help="diagnostic: run a torch CPU sgemm of this edge first (what bench.py's cpu_baseline does)")
    args = ap.parse_args()
    state = {"allowed_cpus_at_start": len(os.sched_getaffinity(0))}
    if args.cpu_warm:
        x = torch.rand(args.cpu_warm, args.cpu_warm)
        torch.mm(x, x)
        del x
        state["allowed_cpus_after_cpu_sgemm"] = len(os.sched_getaffinity(0))
        state["process_threads"] = len(os.listdir("/proc/self/task"))
        state["loadavg"] = os.getloadavg()
    bofhip.require_device()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    # diagnostic pre-stages (what bench.py does before its e2e block), comma separated
    for stage in [x for x in args.pre.split(",") if x]:
        if stage == "alloc":            # allocate and free 16 GiB through torch's allocator
            ts = [torch.empty(1 << 30, dtype=torch.float32, device=dev) for _ in range(4)]
            for t in ts:
                t.zero_()
            torch.cuda.synchronize()
            del ts, t
        elif stage == "resident":       # four passes of the resident cfg2 tile DAG
            n = 32768
            a = torch.empty(n * n, dtype=torch.float32, device=dev)
            b = torch.empty(n * n, dtype=torch.float32, device=dev)
            c = torch.empty(n * n, dtype=torch.float32, device=dev)
            bofhip.gen_dense(a.data_ptr(), 0, n * n, "u", 1, st)
            bofhip.gen_dense(b.data_ptr(), 0, n * n, "u", 2, st)
            for _ in range(4):
                bofhip.gemm_resident("R", "N", "N", n, n, n, 1.0, 0.0, a.data_ptr(), b.data_ptr(), c.data_ptr(), 0, 0, 0,
                                     bofhip.default_options(n_streams=1), st)
            torch.cuda.synchronize()
            del a, b, c
        elif stage == "dgemm":          # a float64 product through torch (rocBLAS)
            x = torch.rand(4096, 8192, dtype=torch.float64, device=dev)
            y = torch.rand(8192, 8192, dtype=torch.float64, device=dev)
            (x @ y).sum().item()
            del x, y
        elif stage == "release":
            bofhip.lib().bof_flash_release()
        torch.cuda.empty_cache()
        state.setdefault("pre", []).append(stage)
    work = tempfile.mkdtemp(prefix="bof_e2e_", dir=args.dir)
    modes = {1: ("odirect",), 0: ("buffered",), -1: ("odirect", "buffered")}[args.direct]
    try:
        out = bench.e2e_gemm(bofhip, torch, dev, st, work, args.n, args.blk, None, args.io_threads, args.reps,
                             modes=modes, gemm_path=args.path, io_chunk_mib=args.chunk_mib, n_streams=args.streams,
                             pinned_slots=args.pinned, hbm_budget=int(args.budget_gib * 2**30),
                             rank_calls=args.rank_calls,
                             **({"devices": [int(x) for x in args.devices.split(",")]} if args.devices else {}))

    finally:
        shutil.rmtree(work, ignore_errors=True)
    out["args"] = vars(args)
    out["process_state"] = state
    if args.budget_gib > 0 and args.path == 1:
        slot = args.blk * args.blk * 4
        out["simulated"] = bofhip.flash_gemm_simulate("R", "N", "N", args.n, args.n, args.n, 0.0, args.blk,
                                                      int(args.budget_gib * 2**30) // slot,
                                                      lookahead=2 * args.pinned)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
