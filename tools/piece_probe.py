#!/usr/bin/env python3
"""GPU box: kernel time of the target workload for several dispatch-piece sizes (demod_set_piece_rows)."""
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
for rnd in range(2):
    for log2rows in (23, 22, 21, 20, 19, 18, 17):
        d.set_piece_rows(1 << log2rows)
        for _ in range(2):
            d.process(iq, out)
        torch.cuda.synchronize(); d.kernel_ms()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            d.process(iq, out)
        e1.record(); torch.cuda.synchronize()
        kms, cnt = d.kernel_ms()
        print(f"piece 2^{log2rows} rows ({(1 << log2rows) >> 10} MiB in): {d.last_call_launches} dispatches, kernels {kms:.3f} ms per pass, "
              f"wall {e0.elapsed_time(e1) / 8:.3f} ms, frac(wall) {4 * (1 << 32) / (e0.elapsed_time(e1) / 8 * 1e-3) / 8e12:.4f}")
