#!/usr/bin/env python3
"""Kernel time of consecutive launches over ~2 s of sustained load (DVFS / thermal behaviour)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import demodulator_amd as dm
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 32
opt = dm.Options(lowpass_in=12500.0, lowpass_out=6500.0)
n = 2 << lg
d = dm.Demodulator(opt); d.enable_timing(True)
iq = torch.empty(n, dtype=torch.uint8, device="cuda"); out = torch.empty(n // 4, dtype=torch.float32, device="cuda")
d.synth_fill(iq, 0, 7); torch.cuda.synchronize()
t0 = time.perf_counter()
group = max(1, int(20e-3 / (n / 2 / 1.4e12)))   # ~20 ms of work per line
while time.perf_counter() - t0 < 2.0:
    for _ in range(group): d.process(iq, out)
    torch.cuda.synchronize()
    kms, k = d.kernel_ms()
    print(f"t={time.perf_counter()-t0:6.3f}s  {k:3d} launches  avg {kms*1e3:8.1f} us  frac {4*(1<<lg)/(kms*1e-3)/8e12:.3f}")
