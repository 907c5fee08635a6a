#!/usr/bin/env python3
"""GPU box: time of a -q1 call (correct_iq_offsets + the table's consumer) at the default chunk size, per library variant.
    python tools/q1_probe.py [log2_bytes=30]     (DEMOD_HIP_LIBRARY picks the variant)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import demodulator_amd as dm

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << lg
iq = torch.empty(n, dtype=torch.uint8, device="cuda")
out = torch.empty(n // 4, dtype=torch.float32, device="cuda")
for name, flags in (("-q 1", dict(mode=0x14)), ("-q 1 -L", dict(mode=0x14, lowpass_in=12500.0, lowpass_out=6500.0))):
    d = dm.Demodulator(dm.Options(**flags))
    d.synth_fill(iq, 0, 7)
    for _ in range(3):
        d.process(iq, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        d.process(iq, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{os.path.basename(os.environ.get('DEMOD_HIP_LIBRARY', 'default')):28s} {name:8s} {ms:8.3f} ms per 2^{lg} bytes  frac {2 * n / (ms * 1e-3) / 8e12:.3f}")
    d.close()
