#!/usr/bin/env python3
"""GPU box: throughput of the -q1 (correctIq) conditioning against the default, device-resident data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import demodulator_amd as dm
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 29   # log2 of the input bytes
n = 1 << lg
iq = torch.empty(n, dtype=torch.uint8, device="cuda"); out = torch.empty(n // 4, dtype=torch.float32, device="cuda")
for mode, name in ((0x10, "-q0 shiftOrigin"), (0x14, "-q1 correctIq"), (0x1C, "-q3 normalizeInput")):
    for B in (262144, 16384):
        d = dm.Demodulator(dm.Options(mode=mode, buf_size=B, lowpass_in=12500.0, lowpass_out=6500.0))
        d.synth_fill(iq, 0, 7)
        for _ in range(2):
            d.process(iq, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            d.process(iq, out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"{name:20s} chunk {B:>7d} B: {ms:8.3f} ms per 2^{lg - 1} samples  {n / 2 / ms / 1e3:10.0f} Msamples/s")
        d.close()
