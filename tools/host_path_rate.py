#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer paths (never the benchmark's `value`)."""
import sys, os, time, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import demodulator_amd as dm, oracle
opt = dm.Options(lowpass_in=12500.0, lowpass_out=6500.0)
n = 1 << 30   # 1 GiB of IQ = 2^29 samples
data = np.frombuffer(np.random.default_rng(1).bytes(n), dtype=np.uint8)
with dm.Demodulator(opt) as d:
    d.process_host(data[: 64 << 20])
    t0 = time.perf_counter(); out = d.process_host(data); dt = time.perf_counter() - t0
print(f"demod_process_host: {n/2/dt/1e6:.0f} Msamples/s ({n/dt/1e9:.2f} GB/s in + {n/dt/1e9:.2f} GB/s out), pageable numpy buffers, 32 MiB pinned pieces")
exe = os.path.join(os.path.dirname(dm.__file__), "demod")
with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
    src, dst = os.path.join(td, "iq.u8"), os.path.join(td, "out.f32")
    data.tofile(src)
    t0 = time.perf_counter(); subprocess.run([exe, "-i", src, "-o", dst, "-L", "12500", "-l", "6500"], check=True); dt = time.perf_counter() - t0
    print(f"demod CLI file->file in /dev/shm, 256 KiB chunks: {n/2/dt/1e6:.0f} Msamples/s ({dt:.2f} s incl. process start and context creation)")
    t0 = time.perf_counter(); subprocess.run([exe, "-i", src, "-o", dst, "-L", "12500", "-l", "6500", "-b", "6"], check=True); dt = time.perf_counter() - t0
    print(f"demod CLI file->file, 16 MiB chunks (-b 6): {n/2/dt/1e6:.0f} Msamples/s ({dt:.2f} s)")
