#!/usr/bin/env python3
"""GPU box: does the relative placement of the input and output buffers matter?
Both streams advance in lockstep (output byte i is written while input byte i is read), so
if the two buffers start at the same offset within the HBM channel/bank interleave, every
read and its write meet in the same bank.  Times the c3 workload with the output buffer
shifted by SKEW bytes inside one allocation."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import demodulator_amd as dm

n_bytes = 2 << 30
opt = dm.Options(lowpass_in=12500.0, lowpass_out=6500.0)
d = dm.Demodulator(opt)
d.enable_timing(True)
iq = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
d.synth_fill(iq, 0, 0xD3300D03)
pad = 64 << 20
big = torch.empty(n_bytes // 4 + pad // 4, dtype=torch.float32, device="cuda")
print("in ptr %x  out base %x" % (iq.data_ptr(), big.data_ptr()))
if len(sys.argv) > 1 and sys.argv[1] == "fine":
    skews = [k * (32 << 10) for k in range(0, 64)] + [k * (256 << 10) for k in range(8, 64)] + [k << 20 for k in range(16, 64, 2)]
else:
    skews = [0, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 20, 2 << 20, 3 << 20, 4 << 20,
             (1 << 20) + 4096, 16 << 20, 0]
t0 = time.time()
while time.time() - t0 < 0.5:
    d.process(iq, big[: n_bytes // 4]); torch.cuda.synchronize()
d.kernel_ms()
for rnd in range(1 if len(sys.argv) > 1 else 2):
    for s in skews:
        out = big[s // 4: s // 4 + n_bytes // 4]
        for _ in range(3):
            d.process(iq, out)
        torch.cuda.synchronize(); d.kernel_ms()
        for _ in range(20):
            d.process(iq, out)
        torch.cuda.synchronize()
        ms, n = d.kernel_ms()
        print(f"round {rnd} skew {s:>9d}  kernel {ms:.4f} ms  frac {4 * (n_bytes // 2) / (ms * 1e-3) / 8e12:.4f}")
