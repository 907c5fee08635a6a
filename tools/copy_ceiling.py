#!/usr/bin/env python3
"""HBM copy ceiling of the box this runs on: read+write bytes per second of a
plain device-to-device copy of the bench's shape (8 GiB in, 8 GiB out), by
torch's copy kernel and by hipMemcpyDtoDAsync.  Context for roofline.frac."""
import time
import torch

n = 1 << 33
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
a.random_(0, 255)
for name, fn in (("torch copy_ (uint8 view as int32x4)", lambda: b.view(torch.int32).copy_(a.view(torch.int32))),
                 ("torch copy_ float32 view", lambda: b.view(torch.float32).copy_(a.view(torch.float32))),
                 ("torch add (read 8G write 8G) f32", lambda: torch.add(a.view(torch.float32), 1.0, out=b.view(torch.float32)))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: {ms:.3f} ms per 8 GiB -> {2 * n / ms / 1e9:.2f} TB/s read+write ({2*n/ms/1e9/8:.3f} of 8 TB/s)")
