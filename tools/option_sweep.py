#!/usr/bin/env python3
"""GPU box: throughput of the hot path over the reference's option space (filter degrees and
types, conditioning, output format) at the default chunk size, device-resident data.  Shows
which builds of demod_rows an option set lands on and what each costs against the default."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import demodulator_amd as dm

n = 2 << 29          # bytes = 2^29 samples
iq = torch.empty(n, dtype=torch.uint8, device="cuda")
out = torch.empty(n // 4, dtype=torch.float32, device="cuda")

L = dict(lowpass_in=12500.0, lowpass_out=6500.0)
cases = [("default (-d 3)", {})]
cases += [(f"-d {d}", dict(out_degree=d)) for d in (1, 2, 4, 5, 6, 7, 8, 9, 12, 16)]
cases += [("-L -l (D = d = 3)", L)]
cases += [(f"-L -D {D} -d {D}", dict(in_degree=D, out_degree=D, **L)) for D in (1, 2, 4, 5, 6, 7, 8, 9, 12, 16)]
cases += [(f"-m {m}" + (" -L" if f else ""), dict(mode=0x10 | m, **f)) for m in (1, 2, 3) for f in ({}, L)]
cases += [(f"-q {q}" + (" -L" if f else ""), dict(mode=0x10 | (q << 2), **f)) for q in (1, 2, 3) for f in ({}, L)]
cases += [("-S 96000 -l 12500", dict(sample_rate=96000.0, lowpass_out=12500.0)),
          ("-S 96000 -L 12500 -l 6500", dict(sample_rate=96000.0, **L))]

for name, flags in cases:
    try:
        d = dm.Demodulator(dm.Options(**flags))
    except Exception as e:  # noqa: BLE001 -- refused designs are part of the picture
        print(f"{name:28s} refused: {str(e)[:90]}")
        continue
    d.synth_fill(iq, 0, 7)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:      # 150 ms of the same work first: a case that starts on a resting chip reads 5-10 % low
        d.process(iq, out)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        d.process(iq, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    taps = f"{len(d.taps('in')[0])}/{len(d.taps('out')[0])}, head {d.head_outputs}"
    print(f"{name:28s} {ms:8.3f} ms per 2^29 samples  {n / 2 / ms / 1e3:10.0f} Msamples/s  frac {2 * n / (ms * 1e-3) / 8e12:.3f}"
          + (f"  taps in/out {taps}" if taps else ""))
    d.close()
