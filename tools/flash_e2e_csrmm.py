#!/usr/bin/env python3
"""cfg3 END TO END through the C ABI (bof_flash_csrmm on the 17.7 GB of files), the measurement and
verification of bench.py's `e2e.csrmm` block with the knobs exposed (BOF_TRACE=1 for the timeline)."""
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
    args = ap.parse_args()
    bofhip.require_device()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    work = tempfile.mkdtemp(prefix="bof_cfg3_", dir=args.dir)
    try:
        out = bench.e2e_csrmm(bofhip, torch, dev, st, work, None, args.io_threads, args.reps, pinned_slots=args.pinned)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
