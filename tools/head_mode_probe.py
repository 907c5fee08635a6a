#!/usr/bin/env python3
"""GPU box: where the time of a process call goes for one library build -- the streaming
kernel alone (the library's HIP events around its launches) against the whole call
(events around the call), per chunk size and head mode.
    [DEMOD_HIP_LIBRARY=...] python tools/head_mode_probe.py [head_mode ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import demodulator_amd as dm
modes = [int(a) for a in sys.argv[1:]] or [-1]
n = 2 << 30
iq = torch.empty(n, dtype=torch.uint8, device="cuda"); out = torch.empty(n // 4, dtype=torch.float32, device="cuda")
tag = os.path.basename(os.environ.get("DEMOD_HIP_LIBRARY", "current"))
for flags in (dict(lowpass_in=12500.0, lowpass_out=6500.0), dict()):
    for B in (16384, 262144):
        for mode in modes:
            d = dm.Demodulator(dm.Options(buf_size=B, **flags))
            d.set_head_mode(mode)
            d.synth_fill(iq, 0, 7)
            d.enable_timing(True)
            for _ in range(20):
                d.process(iq, out)
            torch.cuda.synchronize(); d.kernel_ms()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                d.process(iq, out)
            e1.record(); torch.cuda.synchronize()
            kms, calls = d.kernel_ms()
            print(f"{tag:28s} {'-L -l' if flags else 'default':8s} chunk {B:>7d} mode {mode:2d}: rows kernel {kms:7.4f} ms  whole call {e0.elapsed_time(e1) / 20:7.4f} ms per 2^30 samples")
            d.close()
