#!/usr/bin/env python3
"""Kernel time of demod_rows versus stream length (fixed cost + per-sample cost)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import demodulator_amd as dm

flags = dict(lowpass_in=12500.0, lowpass_out=6500.0) if (len(sys.argv) < 2 or sys.argv[1] == "c3") else {}
opt = dm.Options(**flags)
d = dm.Demodulator(opt)
d.enable_timing(True)
iq = torch.empty(2 << 32, dtype=torch.uint8, device="cuda")
out = torch.empty((2 << 32) // 4, dtype=torch.float32, device="cuda")
d.synth_fill(iq, 0, 1234)
for lg in (24, 26, 27, 28, 29, 30, 31, 32):
    n = 2 << lg
    for _ in range(3):
        d.process(iq[:n], out[: n // 4])
    torch.cuda.synchronize(); d.kernel_ms()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        d.process(iq[:n], out[: n // 4])
    e1.record(); torch.cuda.synchronize()
    kms, _ = d.kernel_ms()
    step = e0.elapsed_time(e1) / 10
    print(f"2^{lg} samples: rows kernel {kms*1e3:9.1f} us  step {step*1e3:9.1f} us  frac {4*(1<<lg)/(kms*1e-3)/8e12:.3f}  ns/Msample {kms*1e6/((1<<lg)/1e6):.3f}")
