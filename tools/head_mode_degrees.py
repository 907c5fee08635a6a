#!/usr/bin/env python3
"""GPU box: where to replay chunk heads, by design (sections per stage) and chunk size: inside the streaming kernel (1),
head kernels (0: thread per chunk up to 2 sections, wavefront per chunk above), automatic (-1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import demodulator_amd as dm
n = 1 << 30
iq = torch.empty(n, dtype=torch.uint8, device="cuda"); out = torch.empty(n // 4, dtype=torch.float32, device="cuda")
L = dict(lowpass_in=12500.0, lowpass_out=6500.0)
for name, kw in (("default", {}), ("-L", L), ("-L -D 5 -d 5", dict(in_degree=5, out_degree=5, **L)), ("-L -D 6 -d 6", dict(in_degree=6, out_degree=6, **L)),
                 ("-L -D 8 -d 8", dict(in_degree=8, out_degree=8, **L)), ("-d 8", dict(out_degree=8))):
    for B in (262144, 65536, 16384):
        row = []
        for mode in (-1, 0, 1):
            d = dm.Demodulator(dm.Options(buf_size=B, **kw))
            d.set_head_mode(mode)
            d.synth_fill(iq, 0, 7)
            for _ in range(3):
                d.process(iq, out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                d.process(iq, out)
            e1.record(); torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / 8)
            d.close()
        print(f"{name:14s} chunk {B:>7d} B: auto {row[0]:.4f}  head kernels {row[1]:.4f}  in the streaming kernel {row[2]:.4f} ms per 2^29 samples")
