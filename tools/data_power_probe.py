#!/usr/bin/env python3
"""GPU box: does the kernel's time depend on the DATA?  Same kernel (BASELINE config 3 / config 2), same sizes, inputs
that toggle more or fewer bits: uniformly random bytes (the benchmark's synthetic stream), a smooth FM-like signal
(what an SDR delivers), a constant byte, zeros.  Interleaved rounds, 150 ms of spin-up per measurement.
    python tools/data_power_probe.py [rounds=3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import demodulator_amd as dm

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = 2 << 30          # bytes = 2^30 samples
out = torch.empty(n // 4, dtype=torch.float32, device="cuda")


def fm_like(nbytes: int) -> torch.Tensor:
    """127.5 + 100 cos/sin(phase), phase = integral of a slow tone: a 2 MiB period repeated (smooth, like SURVEY 8d's secondary input)."""
    m = 1 << 20
    t = np.arange(m, dtype=np.float64)
    phase = 2 * np.pi * (0.05 * t + 40.0 * np.sin(2 * np.pi * t / 4096.0))
    iq = np.empty(2 * m, dtype=np.uint8)
    iq[0::2] = np.clip(np.rint(127.5 + 100.0 * np.cos(phase)), 0, 255)
    iq[1::2] = np.clip(np.rint(127.5 + 100.0 * np.sin(phase)), 0, 255)
    return torch.from_numpy(iq).cuda().repeat(nbytes // (2 * m))


inputs = {}
d0 = dm.Demodulator(dm.Options())
rnd = torch.empty(n, dtype=torch.uint8, device="cuda")
d0.synth_fill(rnd, 0, 0xD3300D03)
d0.close()
inputs["random"] = rnd
inputs["fm_like"] = fm_like(n)
inputs["const_127"] = torch.full((n,), 127, dtype=torch.uint8, device="cuda")
inputs["const_200"] = torch.full((n,), 200, dtype=torch.uint8, device="cuda")
for name, flags in (("c3", dict(lowpass_in=12500.0, lowpass_out=6500.0)), ("c2", dict())):
    d = dm.Demodulator(dm.Options(**flags))
    d.enable_timing(True)
    for r in range(rounds):
        for dname, iq in inputs.items():
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.15:
                d.process(iq, out)
                torch.cuda.synchronize()
            d.kernel_ms()
            for _ in range(20):
                d.process(iq, out)
            torch.cuda.synchronize()
            ms, launches = d.kernel_ms()
            print(f"{r + 1} {name} {dname:10s} {ms:.4f} ms per dispatch  frac {4 * (n // 2) / (ms * 1e-3) / 8e12:.4f}")
    d.close()
