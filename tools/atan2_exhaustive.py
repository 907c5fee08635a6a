#!/usr/bin/env python3
"""GPU box: fast_atan2 (the streaming kernel's atan2) against float64 atan2 over EVERY discriminator argument the
path without -L can produce: zr = a c + b d, zj = b c - a d with a, b, c, d in [-128, 127] are integers of magnitude
<= 32768.  The octant reduction is done with sign-bit masks and a min/max (exact), so the first octant (0 <= y <= x)
carries the approximation error; it is walked completely (5.4e8 pairs), the other octants through the same pairs
with signs and arguments swapped on every 16th column.  Prints the largest error found."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import demodulator_amd as dm

N = 32768
worst = (0.0, None)
with dm.Demodulator(buf_size=4096) as d:
    step = 256
    for x0 in range(0, N + 1, step):
        xs = np.arange(x0, min(x0 + step, N + 1), dtype=np.float32)
        ys = np.arange(0, int(xs[-1]) + 1, dtype=np.float32)
        Y, X = np.meshgrid(ys, xs, indexing="ij")
        keep = Y <= X
        y, x = Y[keep], X[keep]
        variants = [(y, x)]
        if (x0 // step) % 16 == 0:          # the other seven octants
            variants += [(x, y), (x, -y), (y, -x), (-y, -x), (-x, -y), (-x, y), (-y, x)]
        for yy, xx in variants:
            yy = np.ascontiguousarray(yy, dtype=np.float32); xx = np.ascontiguousarray(xx, dtype=np.float32)
            got = d.selftest_atan2(yy, xx).astype(np.float64)
            want = np.arctan2(yy.astype(np.float64), xx.astype(np.float64))
            err = np.abs(got - want)
            i = int(err.argmax())
            if err[i] > worst[0]:
                worst = (float(err[i]), (float(yy[i]), float(xx[i]), float(got[i]), float(want[i])))
        if (x0 // step) % 16 == 0:
            print(f"x up to {int(xs[-1]):5d}: largest |error| so far {worst[0]:.3e} at (y, x, got, want) = {worst[1]}", flush=True)
print(f"fast_atan2 over all integer arguments of magnitude <= {N}: largest |error| {worst[0]:.3e} rad at {worst[1]} "
      f"(half an ulp of pi is 1.2e-7)")
