#!/usr/bin/env python3
"""GPU box: the three forms of the complex stencil (plain / normalized / folded, include/demod_hip.h) on the searched
worst-case inputs of tests/test_gpu_parity.py::adversarial_case, design by design, under the parity criterion.
    python tools/form_compare.py [seeds, default 3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import demodulator_amd as dm
import oracle
from parity import compare_stream
from test_gpu_parity import adversarial_case

orc = oracle.restatement()
fma = oracle.reference_fma() or oracle.restatement_fma()
designs = [dict(lowpass_in=12500.0, lowpass_out=6500.0), dict(sample_rate=96000.0, lowpass_in=12500.0, lowpass_out=6500.0),
           dict(lowpass_in=9600.0), dict(lowpass_in=12500.0, in_degree=5, mode=0x12), dict(lowpass_in=15000.0, lowpass_out=6500.0),
           dict(lowpass_in=8000.0, sample_rate=96000.0), dict(lowpass_in=20000.0, sample_rate=250000.0), dict(lowpass_in=12500.0, in_degree=4)]
for kw in designs:
    info = dm.design(dm.Options(**kw))
    for seed in range(5, 5 + (int(sys.argv[1]) if len(sys.argv) > 1 else 3)):
        opt, data = adversarial_case(kw, seed)
        want = orc.process(oracle.Flags(**opt.__dict__), data)
        line = [f"{kw} seed {seed} (auto: {info.get('stencil_form')}, fold exact: {info['unpack_fold_exact']})"]
        for form in ("plain", "normalized", "folded"):
            with dm.Demodulator(opt) as d:
                if not d.set_stencil_form(form):
                    line.append(f"  {form:10s} not available")
                    continue
                got = d.process_host(data)
            rep = compare_stream(dm, opt, got, want, data)
            line.append(f"  {form:10s} ok={rep.ok} in_band={rep.n_band} cond={rep.n_cond} beyond={rep.n_beyond} wrap={rep.n_wrap} max_abs={rep.max_abs:.3e}")
        if fma is not None:
            rep = compare_stream(dm, opt, fma.process(oracle.Flags(**opt.__dict__), data), want, data)
            line.append(f"  {'ref FMA':10s} ok={rep.ok} in_band={rep.n_band} cond={rep.n_cond} beyond={rep.n_beyond} wrap={rep.n_wrap} max_abs={rep.max_abs:.3e}")
        print("\n".join(line), flush=True)
