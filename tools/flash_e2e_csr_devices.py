#!/usr/bin/env python3
"""cfg3 (flash _csrmm) and cfg5-size (flash _csrgemv) from files through the in-process device list:
bench.py's e2e_csrmm / e2e_csrgemv legs with `devices`.  usage: flash_e2e_csr_devices.py DIR 0,0"""
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

base = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("TMPDIR", "/tmp")
devs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 and sys.argv[2] else None
bofhip.require_device()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
work = tempfile.mkdtemp(prefix="bof_csr_dev_", dir=base)
extra = {"devices": devs} if devs else {}
try:
    out = {"devices": devs,
           "csrmm": bench.e2e_csrmm(bofhip, torch, dev, st, work, None, 8, 2, **extra)}
    bofhip.lib().bof_flash_release()
    out["csrgemv"] = bench.e2e_csrgemv(bofhip, torch, dev, st, work, None, 8, 2, **extra)
finally:
    shutil.rmtree(work, ignore_errors=True)
print(json.dumps(out))
