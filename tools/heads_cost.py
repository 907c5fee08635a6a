import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import demodulator_amd as dm
n = 2 << 29
iq = torch.empty(n, dtype=torch.uint8, device="cuda"); out = torch.empty(n // 4, dtype=torch.float32, device="cuda")
for flags in (dict(lowpass_in=12500.0, lowpass_out=6500.0), dict()):
    for B in (1024, 2048, 4096, 16384, 65536):
        d = dm.Demodulator(dm.Options(buf_size=B, **flags))
        d.synth_fill(iq, 0, 7)
        d.enable_timing(True)
        for _ in range(10):
            d.process(iq, out)
        torch.cuda.synchronize(); d.kernel_ms()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            d.process(iq, out)
        e1.record(); torch.cuda.synchronize()
        kms, calls = d.kernel_ms()
        tot = e0.elapsed_time(e1) / 10
        print(f"{'-L -l' if flags else 'default':8s} chunk {B:>6d}: rows kernel {kms:7.4f} ms, whole call {tot:7.4f} ms -> heads kernel + gaps {tot - kms:7.4f} ms ({(tot-kms)/tot*100:4.1f} %)  [{n // B} heads]")
        d.close()
