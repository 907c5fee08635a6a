#!/usr/bin/env python3
"""Is a 2^32-sample launch slower per sample than 2^30-sample launches because of its
footprint (16 GiB touched) or because of the launch itself?  Same buffers, 1 vs 4 vs 16 launches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import demodulator_amd as dm
opt = dm.Options(lowpass_in=12500.0, lowpass_out=6500.0)
n = 2 << 32
d = dm.Demodulator(opt); d.enable_timing(True)
iq = torch.empty(n, dtype=torch.uint8, device="cuda"); out = torch.empty(n // 4, dtype=torch.float32, device="cuda")
d.synth_fill(iq, 0, 7); torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    d.process(iq, out); torch.cuda.synchronize()
for parts in (1, 4, 16, 1, 4, 16):
    m = n // parts
    d.kernel_ms()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        for k in range(parts):
            d.process(iq[k * m:(k + 1) * m], out[k * m // 4:(k + 1) * m // 4])
    e1.record(); torch.cuda.synchronize()
    kms, cnt = d.kernel_ms()
    print(f"{parts:2d} launches per pass: rows kernels total {kms*parts:.3f} ms per pass, wall {e0.elapsed_time(e1)/5:.3f} ms, frac(kernel) {4*(1<<32)/(kms*parts*1e-3)/8e12:.3f}")
# same 2^30 slice over and over (4 GiB footprint) vs walking through the 16 GiB
m = n // 4
for name, idx in (("same quarter", [0, 0, 0, 0]), ("walking quarters", [0, 1, 2, 3])):
    d.kernel_ms()
    for _ in range(5):
        for k in idx:
            d.process(iq[k * m:(k + 1) * m], out[k * m // 4:(k + 1) * m // 4])
    torch.cuda.synchronize()
    kms, cnt = d.kernel_ms()
    print(f"{name}: {kms*1e3:.1f} us per 2^30 launch, frac {4*(1<<30)/(kms*1e-3)/8e12:.3f}")
