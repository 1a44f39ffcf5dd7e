#!/usr/bin/env python3
"""bench.py's `secondary.kmeans` line alone (flash::kmeans resident, fused store vs the three-call sequence)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
import torch  # noqa: E402
import bofhip  # noqa: E402
import bench  # noqa: E402

bofhip.require_device()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
print(json.dumps(bench.kmeans_secondary(bofhip, torch, dev, st, int(sys.argv[1]) if len(sys.argv) > 1 else 4,
                                        int(sys.argv[2]) if len(sys.argv) > 2 else 4096)))
