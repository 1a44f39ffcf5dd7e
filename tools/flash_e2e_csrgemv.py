#!/usr/bin/env python3
"""cfg5-size csrgemv END TO END through the C ABI (bof_flash_csrgemv on 6.4 GB of files), the measurement
and verification of bench.py's `e2e.csrgemv` block with the knobs exposed (BOF_TRACE=1 for the timeline)."""
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
    ap.add_argument("--io-threads", type=int, default=8)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--pinned", type=int, default=8, help="row blocks in flight (staging contexts)")
    ap.add_argument("--direct", type=int, default=-1, help="1 O_DIRECT only, 0 buffered only, -1 both")
    args = ap.parse_args()
    bofhip.require_device()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    work = tempfile.mkdtemp(prefix="bof_cfg5_", dir=args.dir)
    modes = {1: ("odirect",), 0: ("buffered",)}.get(args.direct, ("odirect", "buffered"))
    try:
        out = bench.e2e_csrgemv(bofhip, torch, dev, st, work, None, args.io_threads, args.reps, modes=modes,
                                pinned_slots=args.pinned)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
