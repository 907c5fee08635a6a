#!/usr/bin/env python3
"""GPU box: replay tests/test_gpu_parity.py::test_random_call_patterns for one seed and print the details of the
first option set whose one-call result leaves the criterion (values, the reference's own FMA build, |z| nearby)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import demodulator_amd as dm
import oracle
from tests import parity
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
orc, fma = oracle.restatement(), oracle.restatement_fma()
rng = np.random.default_rng(seed + 77)
for it in range(rounds):
    with_in = bool(rng.integers(0, 2))
    mode = 0x10 | int(rng.choice([0, 0, 1, 2, 3])) | int(rng.choice([0, 0, 0, 0x8, 0xC]))
    B = int(1 << rng.integers(8, 18))
    opt = dm.Options(sample_rate=float(rng.choice([96000.0, 125000.0])),
                     lowpass_in=float(rng.choice([9600.0, 12500.0])) if with_in else 0.0,
                     lowpass_out=float(rng.choice([6500.0, 12500.0])),
                     in_degree=int(rng.choice([0, 0, int(rng.integers(1, 17))])) if with_in else 0,
                     out_degree=int(rng.choice([3, 3, int(rng.integers(1, 17))])), mode=mode, buf_size=B)
    try:
        des = dm.design(opt)
    except dm.DemodError:
        continue
    unit = 1024 if B >= 1024 else B
    n_units = int(rng.integers(2, max(3, min(6 * B, 1 << 20) // unit)))
    n_bytes = n_units * unit + (int(rng.integers(0, unit)) & ~3 if rng.random() < 0.5 else 0)
    data = rng.integers(0, 256, n_bytes, dtype=np.uint8)
    cuts = rng.integers(1, n_units, size=int(rng.integers(1, 5)))
    s16 = rng.random() < 0.3
    for whole in (True, False):
        rng.choice([-1, 0, 1, 2]); rng.choice([0, 1, 2, 4]); rng.choice([0, 4, 16, 256])
    if s16:
        continue
    with dm.Demodulator(opt) as d:
        got = d.process(torch.from_numpy(data).cuda()).cpu().numpy()
    fl = oracle.Flags(**opt.__dict__)
    want = orc.process(fl, data)
    rep = parity.compare_stream(dm, opt, got, want, data)
    if rep.ok:
        continue
    print("round", it, opt, "n_bytes", n_bytes)
    print(rep)
    wf = fma.process(fl, data) if fma else None
    for i in rep.first_bad[:4]:
        c, n = divmod(i, B // 4)
        print(f" output {i} = chunk {c} output {n}: gpu {got[i]!r} ref {want[i]!r} ref(fma) {None if wf is None else wf[i]!r}")
        print("   neighbours gpu", got[i - 2:i + 3], "\n              ref", want[i - 2:i + 3], "\n          ref fma", None if wf is None else wf[i - 2:i + 3])
        chunk = data[c * B:(c + 1) * B].astype(np.float64)
        x = (chunk.astype(np.int64) - 127).astype(np.int8).astype(np.float64) if (opt.mode >> 2) & 3 == 0 else chunk
        taps, bias = np.asarray(des["taps_in"] if "taps_in" in des else []), 0.0
        print("   design keys", list(des.keys()))
    break
else:
    print("no failing case in", rounds, "rounds")
