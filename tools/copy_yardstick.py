#!/usr/bin/env python3
"""Yardstick for the temperature update's roofline: how fast does this GPU move the same bytes with no stencil at all?
A device-to-device copy of one 256^3 f64 field (134 MB read + 134 MB written, neither resident in the 256 MiB Infinity Cache
when the working set is cycled) and a read-only / write-only pass, timed with HIP events.  GPU box only (uses torch for the
buffers and the copy kernels; not part of the product)."""
import sys
import torch

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = L * L * (L + 4) + 0
dev = torch.device("cuda:0")
# four field-sized buffers cycled so that no copy finds its source or destination in the Infinity Cache
bufs = [torch.rand(n, dtype=torch.float64, device=dev) for _ in range(6)]
def timed(fn, reps=30):
    for _ in range(5):
        fn(0)
    torch.cuda.synchronize()
    ts = []
    for r in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(r); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]
nbytes = n * 8
med, best = timed(lambda r: bufs[(2 * r + 1) % 6].copy_(bufs[(2 * r) % 6]))
print(f"copy   {2 * nbytes / 1e6:.0f} MB moved: median {med:.1f} us  best {best:.1f} us  -> {2 * nbytes / med / 1e6:.2f} TB/s")
med, best = timed(lambda r: bufs[r % 6].sum())
print(f"read   {nbytes / 1e6:.0f} MB: median {med:.1f} us  best {best:.1f} us  -> {nbytes / med / 1e6:.2f} TB/s")
med, best = timed(lambda r: bufs[r % 6].fill_(1.0))
print(f"write  {nbytes / 1e6:.0f} MB: median {med:.1f} us  best {best:.1f} us  -> {nbytes / med / 1e6:.2f} TB/s")
med, best = timed(lambda r: torch.add(bufs[(2 * r) % 6], 1.0, out=bufs[(2 * r + 1) % 6]))
print(f"add    {2 * nbytes / 1e6:.0f} MB moved (x + 1 -> y): median {med:.1f} us  best {best:.1f} us  -> {2 * nbytes / med / 1e6:.2f} TB/s")
