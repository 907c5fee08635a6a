#!/usr/bin/env python3
"""Does the rows kernel's speed depend on how the buffers were allocated? (2^30 samples, c3 flags)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import demodulator_amd as dm
opt = dm.Options(lowpass_in=12500.0, lowpass_out=6500.0)
n = 2 << 30
def run(tag, iq, out, seed):
    d = dm.Demodulator(opt); d.enable_timing(True)
    d.synth_fill(iq, 0, seed)
    for _ in range(3): d.process(iq, out)
    torch.cuda.synchronize(); d.kernel_ms()
    for _ in range(10): d.process(iq, out)
    torch.cuda.synchronize()
    kms, _ = d.kernel_ms()
    print(f"{tag}: {kms*1e3:.1f} us frac {4*(1<<30)/(kms*1e-3)/8e12:.3f} iq@{iq.data_ptr():#x} out@{out.data_ptr():#x}")
    d.close()
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n // 4, dtype=torch.float32, device="cuda")
run("exact-size fresh allocations", a, b, 0xD3300D03)
run("same, seed 1234", a, b, 1234)
del a, b; torch.cuda.empty_cache()
big_i = torch.empty(2 << 32, dtype=torch.uint8, device="cuda"); big_o = torch.empty((2 << 32) // 4, dtype=torch.float32, device="cuda")
run("slices of 8 GiB buffers", big_i[:n], big_o[: n // 4], 0xD3300D03)
run("slices at +4 GiB offset", big_i[2 * n: 3 * n], big_o[n // 2: n // 2 + n // 4], 0xD3300D03)
del big_i, big_o; torch.cuda.empty_cache()
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n // 4, dtype=torch.float32, device="cuda")
run("exact-size after freeing the big ones", a, b, 0xD3300D03)
