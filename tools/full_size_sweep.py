#!/usr/bin/env python3
"""GPU box: EVERY output of 2^29 samples against the oracle (tests/parity.py's criterion, counts printed) for
designs other than the BASELINE ones -- what the full-size tests do for configs 2 and 3, as a one-off sweep.
    python tools/full_size_sweep.py [log2 of the input bytes, default 30] [seed] [only cases whose name contains this]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import demodulator_amd as dm
import oracle
from parity import compare_stream

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0x5EED
orc = oracle.restatement()
threads = max(1, min(len(os.sched_getaffinity(0)), 64))
L = dict(lowpass_in=12500.0, lowpass_out=6500.0)
cases = [("C2 (defaults)", {}), ("C3 / target (-L 12500 -l 6500)", L), ("-S 96000 -L 12500 -l 6500 (README)", dict(sample_rate=96000.0, **L)), ("-L 9600", dict(lowpass_in=9600.0)),
         ("-S 96000 -l 12500 (README, no -L)", dict(sample_rate=96000.0, lowpass_out=12500.0)),
         ("-m 3 -L 12500 -l 6500 (Chebyshev)", dict(mode=0x13, **L)), ("-L -D 6 -d 6 (3 taps)", dict(in_degree=6, out_degree=6, **L)),
         ("-q 3 -L", dict(mode=0x1C, **L)), ("-q 2", dict(mode=0x18)), ("-L -D 9 -d 9", dict(in_degree=9, out_degree=9, **L)),
         ("-b -4 -L (16 KiB chunks)", dict(buf_size=16384, **L)), ("-b -8 -L (1 KiB chunks)", dict(buf_size=1024, **L)),
         ("-q 1", dict(mode=0x14)), ("-q 1 -L", dict(mode=0x14, **L)), ("-q 1 -b -6 -S 192000", dict(mode=0x14, buf_size=4096, sample_rate=192000.0)),
         ("-b -10 (256-byte chunks)", dict(buf_size=256)), ("-b -10 -L", dict(buf_size=256, **L))]
only = sys.argv[3] if len(sys.argv) > 3 else ""
cases = [c for c in cases if only in c[0]]
n = 1 << lg
iq = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, kw in cases:
    opt = dm.Options(**kw)
    B = opt.buf_size
    fl = oracle.Flags(**opt.__dict__)
    with dm.Demodulator(opt) as d:
        d.synth_fill(iq, 0, seed)
        out = torch.full((n // 4,), float("nan"), dtype=torch.float32, device="cuda")
        d.process(iq, out)
        torch.cuda.synchronize()
        assert not bool(torch.isnan(out).any())
    total = None
    # chunks are independent, so a slab is a stream of its own for the oracle -- except with -q1, whose
    # running offsets carry through the whole stream: one piece there
    slab = n if (opt.mode >> 2) & 3 == 1 else max(B, 64 << 20)
    for o in range(0, n, slab):
        data = iq[o:o + slab].cpu().numpy()
        got = out[o // 4:(o + slab) // 4].cpu().numpy()
        want = orc.process(fl, data, threads=threads)
        rep = compare_stream(dm, opt, got, want, data)
        total = rep if total is None else total + rep
    print(f"{name:38s} {'ok  ' if total.ok else 'FAIL'} {total}", flush=True)
