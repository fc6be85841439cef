#!/usr/bin/env python3
"""Probe: device-to-host copy rate of a 947 MB field into (a) pinned memory, (b) registered pageable memory,
with the copy issued on one stream or split over two / four streams (do the copies of different streams use
different copy engines and add up on the PCIe link?)."""
import time
import numpy as np
import torch

n = 118425857
src = torch.rand(n, dtype=torch.float64, device="cuda")
pinned = torch.empty(n, dtype=torch.float64).pin_memory()
page = np.empty(n, dtype=np.float64)
page[:] = 0.0
rt = torch.cuda.cudart()
assert int(rt.cudaHostRegister(page.ctypes.data, page.nbytes, 0)) == 0
reg = torch.from_numpy(page)


def run(dst, k):
    streams = [torch.cuda.Stream() for _ in range(k)]
    cuts = np.linspace(0, n, k + 1).astype(np.int64)
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                dst[cuts[i]:cuts[i + 1]].copy_(src[cuts[i]:cuts[i + 1]], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return n * 8 / best / 1e9, best * 1e3


for name, dst in (("pinned", pinned), ("registered pageable", reg)):
    for k in (1, 2, 4):
        gbs, ms = run(dst, k)
        print("%-20s %d stream(s): %6.1f GB/s  %6.2f ms" % (name, k, gbs, ms), flush=True)
