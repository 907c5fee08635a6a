#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer paths (never the benchmark's `value`), next to what
the box's host memory, PCIe link and /dev/shm do on their own."""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import demodulator_amd as dm

opt = dm.Options(lowpass_in=12500.0, lowpass_out=6500.0)
n = 1 << 30   # 1 GiB of IQ = 2^29 samples
data = np.frombuffer(np.random.default_rng(1).bytes(n), dtype=np.uint8)


def best(fn, reps=3):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return min(t)


# -- the box on its own ------------------------------------------------------------------
dst = np.empty_like(data)
dt = best(lambda: np.copyto(dst, data))
print(f"host memcpy, one thread:                 {n / dt / 1e9:6.2f} GB/s")
pin = torch.from_numpy(data.copy()).pin_memory()
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
back = torch.empty(n, dtype=torch.uint8).pin_memory()
def h2d():
    dev.copy_(pin, non_blocking=True); torch.cuda.synchronize()
def d2h():
    back.copy_(dev, non_blocking=True); torch.cuda.synchronize()
print(f"PCIe pinned -> device (hipMemcpyAsync):  {n / best(h2d) / 1e9:6.2f} GB/s")
print(f"PCIe device -> pinned:                   {n / best(d2h) / 1e9:6.2f} GB/s")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1):
        dev.copy_(pin, non_blocking=True)
    with torch.cuda.stream(s2):
        back.copy_(dev2, non_blocking=True)
    torch.cuda.synchronize()
dev2 = torch.empty(n, dtype=torch.uint8, device="cuda")
print(f"PCIe both directions at once:            {n / best(both) / 1e9:6.2f} GB/s each way")
del dev, dev2, back
with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
    f = os.path.join(td, "x")
    def wr():
        with open(f, "wb", buffering=0) as fh:
            for o in range(0, n, 8 << 20):
                fh.write(data[o:o + (8 << 20)])
    def rd():
        with open(f, "rb", buffering=0) as fh:
            while fh.readinto(memoryview(dst)[: 8 << 20]):
                pass
    print(f"/dev/shm write(), one thread, new file:  {n / best(wr, 2) / 1e9:6.2f} GB/s")
    print(f"/dev/shm write(), one thread, rewrite:   {n / best(wr, 2) / 1e9:6.2f} GB/s")
    print(f"/dev/shm read(), one thread:             {n / best(rd, 2) / 1e9:6.2f} GB/s")

# -- demod_process_host ---------------------------------------------------------------------
with dm.Demodulator(opt) as d:
    span = dm._native.Span(0, None, 1, 0)
    page_out = np.zeros(n // 4, dtype=np.float32)     # touched once: page faults are not the path's
    def pageable():
        d._check(d._lib.demod_process_host(d._ctx, data.ctypes.data, n, page_out.ctypes.data, C.byref(span)), "process_host")
    pageable()
    dt = best(pageable)
    print(f"demod_process_host, pageable buffers:    {n / 2 / dt / 1e6:8.0f} Msamples/s = {n / dt / 1e9:.2f} GB/s in + {n / dt / 1e9:.2f} GB/s out")
    pout = torch.empty(n // 4, dtype=torch.float32).pin_memory()
    def pinned():
        d._check(d._lib.demod_process_host(d._ctx, pin.data_ptr(), n, pout.data_ptr(), C.byref(span)), "process_host")
    pinned()
    dt = best(pinned)
    print(f"demod_process_host, pinned buffers:      {n / 2 / dt / 1e6:8.0f} Msamples/s = {n / dt / 1e9:.2f} GB/s in + {n / dt / 1e9:.2f} GB/s out")

# -- the CLI ------------------------------------------------------------------------------------
# steady-state rate = extra bytes / extra time between a 256 MiB and a 4 GiB input (process
# start, HIP initialisation and context creation are in both)
exe = os.path.join(os.path.dirname(dm.__file__), "demod")
with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
    small, big, out = os.path.join(td, "small.u8"), os.path.join(td, "big.u8"), os.path.join(td, "out.f32")
    data[: 256 << 20].tofile(small)
    with open(big, "wb") as fh:
        for _ in range(4):
            fh.write(data)
    nb, ns = 4 * n, 256 << 20
    def run(src, target, extra):
        t0 = time.perf_counter()
        subprocess.run([exe, "-i", src, "-o", target, "-L", "12500", "-l", "6500"] + extra, check=True,
                       env={**os.environ, "DEMOD_RING_STATS": "1"})
        return time.perf_counter() - t0
    for label, extra in (("256 KiB chunks (default)", []), ("16 MiB chunks (-b 6)", ["-b", "6"]), ("16 KiB chunks (-b -4)", ["-b", "-4"])):
        for target in (out, "/dev/null"):
            ts = min(run(small, target, extra) for _ in range(2))
            tb = min(run(big, target, extra) for _ in range(2))
            print(f"demod CLI file -> {'file in /dev/shm' if target == out else '/dev/null       '}, {label:24s}: "
                  f"{(nb - ns) / 2 / (tb - ts) / 1e6:6.0f} Msamples/s steady state  (4 GiB in {tb:.2f} s, 256 MiB in {ts:.2f} s)")
            if target == out:
                os.unlink(out)
