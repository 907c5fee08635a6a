#!/usr/bin/env python3
"""GPU box: throughput vs chunk size (-b) -- every chunk has a head to replay.
    python tools/chunk_sweep.py [head_mode]     head_mode: -1 automatic (default), 0 demod_heads kernel, 1 in the streaming kernel"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import demodulator_amd as dm
mode = int(sys.argv[1]) if len(sys.argv) > 1 else -1
n = 2 << 29
iq = torch.empty(n, dtype=torch.uint8, device="cuda"); out = torch.empty(n // 4, dtype=torch.float32, device="cuda")
for flags in (dict(), dict(lowpass_in=12500.0, lowpass_out=6500.0)):
    for B in (256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 4 << 20):
        d = dm.Demodulator(dm.Options(buf_size=B, **flags))
        d.set_head_mode(mode)
        d.synth_fill(iq, 0, 7)
        # spin-up: after a slow, light kernel (the 256-byte case takes 6-16 ms per call) the device needs tens of
        # milliseconds of load to be back at the clocks it holds under it; three calls were not enough (round 3: the
        # 1-2 KiB cases read 5-10 % low)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.15:
            d.process(iq, out)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            d.process(iq, out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{'-L -l' if flags else 'default':8s} chunk {B:>8d} B: {ms:8.3f} ms per 2^29 samples  {n / 2 / ms / 1e3:10.0f} Msamples/s  frac {2 * n / (ms * 1e-3) / 8e12:.3f}")
        d.close()
