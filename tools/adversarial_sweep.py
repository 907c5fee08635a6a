#!/usr/bin/env python3
"""GPU box: tests/test_gpu_parity.py::test_adversarial_small_filtered_samples over RANDOM designs -- for every design the
1,024 sample groups with the smallest |z| there are (2- and 3-tap complex stencils, -q0 folded or not, -q3), planted among
random neighbours, against the oracle under the criterion.  Prints the designs that leave it (none expected) and totals.
    python tools/adversarial_sweep.py [designs, default 300] [seed] [fold: ask for the folded unpack where a design has it]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import demodulator_amd as dm
import oracle
from parity import compare_stream

n_designs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc, fma = oracle.restatement(), oracle.restatement_fma()
ask_fold = len(sys.argv) > 3 and sys.argv[3] == "fold"
done = bad = bad_self = folded = 0
tot = dict(n=0, cond=0, wrap=0, band=0)
worst = 0.0
while done < n_designs:
    opt = dm.Options(buf_size=65536, sample_rate=float(rng.choice([96000.0, 125000.0, 192000.0, 250000.0])),
                     lowpass_in=float(rng.integers(3000, 30000)), lowpass_out=float(rng.choice([6500.0, 12500.0])),
                     in_degree=int(rng.integers(1, 9)), out_degree=int(rng.choice([3, 3, 5, 7])),
                     mode=0x10 | int(rng.choice([0, 0, 1, 2, 3])) | int(rng.choice([0, 0, 0, 0xC])))
    try:
        din = dm.design(opt, "in")
    except dm.DemodError:
        continue
    h = np.asarray(din["taps"], dtype=np.float64)
    T = h.size
    if T not in (2, 3):
        continue
    q3 = (opt.mode >> 2) & 3 == 3
    v = np.arange(256, dtype=np.float64) * (2.0 / 255.0) - 1.0 if q3 else np.arange(-128, 128, dtype=np.float64)
    to_byte = (lambda i: i) if q3 else (lambda i: (i - 128 + 127) & 255)
    lane = h[0] * v[:, None] + h[1] * v[None, :]
    if T == 3:
        lane = lane[:, :, None] + h[2] * v[None, None, :]
    best = []
    for bias in (0.0, float(din["bias_im"])):
        flat = np.argpartition(np.abs(lane + bias), 32, axis=None)[:32]
        best.append([np.unravel_index(int(i), lane.shape) for i in flat])
    groups = [(re, im) for re in best[0] for im in best[1]]
    data = rng.integers(0, 256, 4 * opt.buf_size, dtype=np.uint8)
    slots = rng.permutation(np.arange(40, 4 * opt.buf_size // 2 - 40, 7))[:len(groups)]
    for k, (re, im) in zip(slots, groups):
        for j in range(T):
            data[2 * (k + j)] = to_byte(int(re[j]))
            data[2 * (k + j) + 1] = to_byte(int(im[j]))
    with dm.Demodulator(opt) as d:
        is_folded = d.set_unpack_fold(True) if ask_fold else False
        got = d.process_host(data)
    want = orc.process(oracle.Flags(**opt.__dict__), data)
    rep = compare_stream(dm, opt, got, want, data)
    done += 1
    folded += is_folded
    tot["n"] += rep.n; tot["cond"] += rep.n_cond; tot["wrap"] += rep.n_wrap; tot["band"] += rep.n_band
    if not rep.ok:
        # the reference's own FMA build on the same bytes: outside the criterion as well (then the criterion's model
        # does not reach this far: |z| of 1e-5 and angles that are noise), or only this path?
        self_rep = compare_stream(dm, opt, fma.process(oracle.Flags(**opt.__dict__), data), want, data) if fma else None
        both = self_rep is not None and not self_rep.ok
        bad += 1
        bad_self += both
        zmin = min(abs(lane + 0.0).min(), abs(lane + float(din["bias_im"])).min())
        print(f"OUTSIDE THE CRITERION ({'the reference build pair too' if both else 'THIS PATH ONLY'}; {T} taps, "
              f"{'folded' if is_folded else 'unfolded'} unpack, smallest lane value {zmin:.2e}): -S {opt.sample_rate:.0f} -L {opt.lowpass_in:.0f} "
              f"-D {opt.in_degree} -d {opt.out_degree} mode {opt.mode:#x}: {rep}" + (f" | reference pair: {self_rep}" if self_rep else ""), flush=True)
        if os.environ.get("DEMOD_ADV_VERBOSE") == "1":
            import parity
            c = parity.conditioning(dm, opt, data)
            taps = [float(t) for t in dm.design(opt)["taps"]]
            for i in rep.first_bad[:4]:
                print(f"    output {i}: this path {got[i]!r} reference {want[i]!r} difference {got[i] - want[i]:+.5f} = {(got[i] - want[i]) / (2 * np.pi):+.4f} x 2 pi; "
                      f"clause {c.cond[i]:.4f}, near the axis {bool(c.near_axis[i])}, output taps {taps}", flush=True)
    if done % 50 == 0:
        print(f"{done} designs ({folded} with the folded unpack), {bad} outside the criterion; outputs {tot['n']}, needed the clause {tot['cond']}, wraps {tot['wrap']}, in band {tot['band']}", flush=True)
print(f"adversarial sweep: {done} designs ({folded} folded), {bad} outside the criterion ({bad_self} of them with the reference's own build pair outside it too); outputs {tot['n']}, needed the clause {tot['cond']}, wraps {tot['wrap']}, in band {tot['band']}")
