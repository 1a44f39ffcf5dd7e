#!/usr/bin/env python3
"""How fast can the chip do random 4-byte gathers from a 142 MB / 200 MB vector (what bounds csrgemv 'N')?
torch.index_select as an independent kernel: 5e8 int64 indices (4 GB, streamed) -> 5e8 gathered floats."""
import torch
dev = torch.device("cuda:0")
for live in (35_500_000, 50_000_000, 4_000_000, 500_000):
    x = torch.arange(50_000_000, device=dev, dtype=torch.float32)
    idx = torch.randint(0, live, (500_000_000,), device=dev, dtype=torch.int64)
    for int32 in (False, True):
        ii = idx.int() if int32 else idx
        torch.index_select(x, 0, ii[:1000])
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            y = torch.index_select(x, 0, ii)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"live {live * 4 / 1e6:.0f} MB, {'int32' if int32 else 'int64'} indices: {best:.2f} ms = {5e8 / best / 1e6:.1f} G gathers/s", flush=True)
        del y
    del idx, ii
