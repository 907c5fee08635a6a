#!/usr/bin/env python3
"""GPU box: throughput of the whole-chunk literal kernel (chunks below one row, -b -9 ... -b -16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import demodulator_amd as dm
n = 1 << 28
iq = torch.empty(n, dtype=torch.uint8, device="cuda"); out = torch.empty(n // 4, dtype=torch.float32, device="cuda")
for flags in (dict(), dict(lowpass_in=12500.0, lowpass_out=6500.0)):
    for B in (512, 256, 64, 16, 4):
        d = dm.Demodulator(dm.Options(buf_size=B, **flags))
        d.synth_fill(iq, 0, 7)
        d.process(iq, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            d.process(iq, out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 2
        print(f"{'-L -l' if flags else 'default':8s} chunk {B:>4d} B: {ms:8.2f} ms per 2^27 samples  {n / 2 / ms / 1e3:9.0f} Msamples/s")
        d.close()
