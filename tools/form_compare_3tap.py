import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import demodulator_amd as dm
import oracle
from parity import compare_stream
from test_gpu_parity import adversarial_case
orc = oracle.restatement()
designs = [dict(sample_rate=192000.0, lowpass_in=6265.0, in_degree=8, out_degree=3), dict(sample_rate=250000.0, lowpass_in=26450.0, in_degree=8, out_degree=7),
           dict(sample_rate=96000.0, lowpass_in=27012.0, in_degree=6, out_degree=5, mode=0x11), dict(sample_rate=125000.0, lowpass_in=10039.0, in_degree=6, out_degree=3, mode=0x12),
           dict(sample_rate=250000.0, lowpass_in=26283.0, in_degree=7, out_degree=5, mode=0x11), dict(sample_rate=96000.0, lowpass_in=15541.0, in_degree=8, out_degree=3, mode=0x11),
           dict(sample_rate=96000.0, lowpass_in=21031.0, in_degree=7, out_degree=3, mode=0x12), dict(sample_rate=192000.0, lowpass_in=26926.0, in_degree=2, out_degree=3, mode=0x12),
           dict(sample_rate=125000.0, lowpass_in=21359.0, in_degree=4, out_degree=3), dict(lowpass_in=12500.0, in_degree=6), dict(lowpass_in=9600.0, in_degree=8, out_degree=7, sample_rate=96000.0)]
tot = {}
for kw in designs:
    info = dm.design(dm.Options(**kw))
    for seed in (5, 6, 7):
        opt, data = adversarial_case(kw, seed)
        want = orc.process(oracle.Flags(**opt.__dict__), data)
        line = [f"{kw} seed {seed} (auto: {info.get('stencil_form')}; norm err {info.get('normalized_error', 0):.2e} plain {info.get('plain_error', 0):.2e})"]
        for form in ("plain", "normalized"):
            with dm.Demodulator(opt) as d:
                if not d.set_stencil_form(form):
                    line.append(f"  {form:10s} not available"); continue
                got = d.process_host(data)
            rep = compare_stream(dm, opt, got, want, data)
            t = tot.setdefault(form, [0, 0, 0, 0]); t[0] += rep.n_band; t[1] += rep.n_cond; t[2] += rep.n_beyond; t[3] += (not rep.ok)
            line.append(f"  {form:10s} ok={rep.ok} in_band={rep.n_band} cond={rep.n_cond} beyond={rep.n_beyond} wrap={rep.n_wrap} max_abs={rep.max_abs:.3e}")
        print("\n".join(line), flush=True)
print(tot)
