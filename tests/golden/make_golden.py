#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (oracle/_ref, compiled from
/root/reference by oracle/Makefile).  Run in the build container only:

    python tests/golden/make_golden.py

Fixtures are data: PRNG-defined inputs (regenerated from the seed, not stored,
except where noted) and the float32 outputs the reference's own functions
produced for them.  The reference publishes no golden vectors of its own
(SURVEY.md section 4), so these are the pins for the oracle and the HIP path.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402

DEFAULT_B = 262144


def flagset(**kw) -> dict:
    d = dict(sample_rate=125000.0, lowpass_in=0.0, lowpass_out=12500.0, epsilon=0.3,
             in_degree=0, out_degree=3, mode=0x10, buf_size=DEFAULT_B)
    d.update(kw)
    return d


# name -> (flags, seed, n_chunks).  Inputs: oracle.synth_bytes(seed, 0, n_chunks*buf_size)
STREAM_CASES = {
    # BASELINE config 2 / 3 flag sets at tiny and default chunk sizes
    "c2_b1024": (flagset(buf_size=1024), 0xD3300D02, 8),
    "c3_b1024": (flagset(lowpass_in=12500.0, lowpass_out=6500.0, buf_size=1024), 0xD3300D03, 8),
    "c2_default": (flagset(), 0xD3300D02, 2),
    "c3_default": (flagset(lowpass_in=12500.0, lowpass_out=6500.0), 0xD3300D03, 2),
    # README pipes (README.md:19-31)
    "readme_dsd_b16384": (flagset(sample_rate=96000.0, lowpass_out=12500.0, buf_size=16384),
                          0xD3300D11, 2),
    "readme_multimon_b16384": (flagset(sample_rate=96000.0, lowpass_out=6500.0,
                                       lowpass_in=12500.0, buf_size=16384), 0xD3300D12, 2),
    # test.sh:70,109-124 option sets: -m3 (both Chebyshev), -d5 -L12500
    "testsh_m3_b4096": (flagset(sample_rate=96000.0, lowpass_out=12500.0, mode=0x13,
                                buf_size=4096), 0xD3300D13, 2),
    "testsh_m3_L_b4096": (flagset(sample_rate=96000.0, lowpass_out=9600.0, lowpass_in=12500.0,
                                  mode=0x13, buf_size=4096), 0xD3300D14, 2),
    "testsh_d5_L_b4096": (flagset(sample_rate=96000.0, lowpass_out=6500.0, lowpass_in=12500.0,
                                  out_degree=5, buf_size=4096), 0xD3300D15, 2),
    # -q3 normalizeInput conditioning (src/matrix.c:100-118)
    "q3_b4096": (flagset(mode=0x1C, buf_size=4096), 0xD3300D16, 2),
    "q3_L_b4096": (flagset(mode=0x1C, lowpass_in=12500.0, lowpass_out=6500.0, buf_size=4096),
                   0xD3300D17, 2),
    # tiny chunks: one 1 KiB row and below (-b -10 .. -b -16)
    "c3_b256": (flagset(lowpass_in=12500.0, lowpass_out=6500.0, buf_size=256), 0xD3300D18, 4),
    "c2_b64": (flagset(buf_size=64), 0xD3300D19, 4),
    "c3_b32": (flagset(lowpass_in=12500.0, lowpass_out=6500.0, buf_size=32), 0xD3300D1A, 4),
    "c2_b4": (flagset(buf_size=4), 0xD3300D1B, 16),
}
# every output / input degree 1..8 (1-3 tap stencils, imag bias -1/-2/-5/-10)
for _d in range(1, 9):
    STREAM_CASES[f"deg{_d}_b4096"] = (flagset(out_degree=_d, buf_size=4096), 0xD3300D20 + _d, 2)
    STREAM_CASES[f"deg{_d}_L_b4096"] = (
        flagset(out_degree=_d, lowpass_in=12500.0, lowpass_out=6500.0, buf_size=4096),
        0xD3300D30 + _d, 2)
STREAM_CASES["D6_d3_L_b4096"] = (flagset(in_degree=6, out_degree=3, lowpass_in=12500.0,
                                         lowpass_out=6500.0, buf_size=4096), 0xD3300D40, 2)
STREAM_CASES["D2_d7_L_b4096"] = (flagset(in_degree=2, out_degree=7, lowpass_in=12500.0,
                                         lowpass_out=6500.0, buf_size=4096), 0xD3300D41, 2)

# -q2 highpassDc conditioning (src/matrix.c:142-157).  Chunks of 32 KiB and more only:
# below that the reference reads heap contents behind highpassDc's buffer (undefined).
# The reference runs these in a process of its own (oracle/ref_stream_main.c).
STREAM_CASES["q2_b32768"] = (flagset(mode=0x18, buf_size=32768), 0xD3300D50, 2)
STREAM_CASES["q2_d5_S96k_b65536"] = (flagset(mode=0x18, sample_rate=96000.0, out_degree=5,
                                             buf_size=65536), 0xD3300D51, 2)
STREAM_CASES["q2_L_b32768"] = (flagset(mode=0x18, lowpass_in=12500.0, lowpass_out=6500.0,
                                       buf_size=32768), 0xD3300D52, 2)

# -q1 correctIq conditioning (src/matrix.c:120-140): running DC offsets, carried from chunk to
# chunk (static state: the reference runs each stream in a process of its own).  Small chunks
# are the interesting ones: there the state a chunk starts from decides its results.
STREAM_CASES["q1_b32768"] = (flagset(mode=0x14, buf_size=32768), 0xD3300D60, 4)
STREAM_CASES["q1_L_b16384"] = (flagset(mode=0x14, lowpass_in=12500.0, lowpass_out=6500.0,
                                       buf_size=16384), 0xD3300D61, 4)
STREAM_CASES["q1_b1024"] = (flagset(mode=0x14, buf_size=1024), 0xD3300D62, 16)
STREAM_CASES["q1_S96k_d5_b4096"] = (flagset(mode=0x14, sample_rate=96000.0, out_degree=5,
                                            buf_size=4096), 0xD3300D63, 6)

# known-answer byte quads, SURVEY.md Appendix A.2 (signed zeros, +-pi, wrap of 255)
QUADS = np.array([
    [127, 127, 122, 124], [127, 127, 132, 130], [127, 127, 127, 127], [127, 130, 127, 120],
    [130, 127, 120, 127], [120, 127, 130, 127], [255, 255, 0, 0], [0, 0, 255, 255],
    [128, 127, 126, 127], [126, 127, 128, 127], [127, 126, 127, 128], [127, 128, 127, 126],
    [255, 127, 0, 127], [0, 127, 255, 127], [127, 255, 127, 0], [127, 0, 127, 255],
    [128, 128, 126, 126], [126, 126, 128, 128], [200, 50, 50, 200], [1, 254, 254, 1],
], dtype=np.uint8)

SOS_CASES = [(kind, deg, fc, fs, eps)
             for kind in (0, 1, 2, 3)
             for deg in range(1, 9)
             for (fc, fs) in ((12500.0, 125000.0), (6500.0, 125000.0), (12500.0, 96000.0),
                              (6500.0, 96000.0), (9600.0, 96000.0), (1.0, 125000.0))
             for eps in (0.3,)] + [(1, 3, 12500.0, 96000.0, 0.1), (1, 4, 12500.0, 96000.0, 0.2)]


def fm_like_bytes(n_samples: int) -> np.ndarray:
    """Host-generated FM-like IQ (secondary input): a
    sinusoidally modulated carrier plus a little noise, 127.5 + 100*cos/sin,
    rounded to uint8.  Stored in the fixture (cos/sin are not bit-portable)."""
    rng = np.random.default_rng(0xF00D)
    t = np.arange(n_samples, dtype=np.float64) / 192000.0
    phase = 2 * np.pi * 3000.0 * t + 5.0 * np.sin(2 * np.pi * 440.0 * t)
    i = 127.5 + 100.0 * np.cos(phase) + rng.normal(0, 2.0, n_samples)
    q = 127.5 + 100.0 * np.sin(phase) + rng.normal(0, 2.0, n_samples)
    iq = np.empty(2 * n_samples, dtype=np.float64)
    iq[0::2], iq[1::2] = i, q
    return np.clip(np.rint(iq), 0, 255).astype(np.uint8)


def main() -> None:
    ref = oracle.reference()
    if ref is None:
        raise SystemExit("oracle/_ref is not built (needs /root/reference): run `make -C oracle`")

    manifest = {"generator": "tests/golden/make_golden.py",
                "source": "reference functions compiled from /root/reference (oracle/_ref)",
                "input_rule": "oracle.synth_bytes(seed, 0, n_chunks*buf_size)",
                "streams": {}, "files": []}

    for name, (flags, seed, n_chunks) in STREAM_CASES.items():
        fl = oracle.Flags(**flags)
        data = oracle.synth_bytes(seed, 0, n_chunks * fl.buf_size)
        out = ref.process(fl, data)
        assert out.size == n_chunks * fl.buf_size // 4
        fn = f"stream_{name}.npy"
        np.save(os.path.join(HERE, fn), out)
        manifest["streams"][name] = {"flags": flags, "seed": seed, "n_chunks": n_chunks,
                                     "file": fn}
        manifest["files"].append(fn)

    # C1 plumbing input: FM-like signal through the README dsd flag set
    fm = fm_like_bytes(49152)  # 98304 bytes = 6 chunks of 16384
    fl = oracle.Flags(**flagset(sample_rate=96000.0, lowpass_out=12500.0, buf_size=16384))
    np.save(os.path.join(HERE, "fm_like_input.npy"), fm)
    np.save(os.path.join(HERE, "stream_fm_like_dsd.npy"), ref.process(fl, fm))
    fl2 = oracle.Flags(**flagset(sample_rate=96000.0, lowpass_out=6500.0, lowpass_in=12500.0,
                                 buf_size=16384))
    np.save(os.path.join(HERE, "stream_fm_like_multimon.npy"), ref.process(fl2, fm))
    manifest["fm_like"] = {"input": "fm_like_input.npy",
                           "dsd": {"flags": flagset(sample_rate=96000.0, lowpass_out=12500.0,
                                                    buf_size=16384),
                                   "file": "stream_fm_like_dsd.npy"},
                           "multimon": {"flags": flagset(sample_rate=96000.0, lowpass_out=6500.0,
                                                         lowpass_in=12500.0, buf_size=16384),
                                        "file": "stream_fm_like_multimon.npy"}}
    manifest["files"] += ["fm_like_input.npy", "stream_fm_like_dsd.npy",
                          "stream_fm_like_multimon.npy"]

    # primitives: unpack of every byte value, discriminator known answers
    allbytes = np.arange(256, dtype=np.uint8)
    np.save(os.path.join(HERE, "unpack_shift_0_255.npy"), ref.unpack_shift(allbytes))
    np.save(os.path.join(HERE, "unpack_normalize_0_255.npy"), ref.unpack_normalize(allbytes))
    np.save(os.path.join(HERE, "quads_bytes.npy"), QUADS)
    np.save(os.path.join(HERE, "quads_discriminator.npy"),
            ref.discriminate(ref.unpack_shift(QUADS.reshape(-1))))
    manifest["files"] += ["unpack_shift_0_255.npy", "unpack_normalize_0_255.npy",
                          "quads_bytes.npy", "quads_discriminator.npy"]

    # designed SOS rows, hex-exact
    sos_tab = []
    for (kind, deg, fc, fs, eps) in SOS_CASES:
        sos = ref.design_sos(kind, deg, fc, fs, eps)
        sos_tab.append({"kind": kind, "degree": deg, "fc": fc, "fs": fs, "epsilon": eps,
                        "sos_hex": [[float(v).hex() for v in row] for row in sos]})
    manifest["sos"] = sos_tab

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    total = sum(os.path.getsize(os.path.join(HERE, fn)) for fn in manifest["files"])
    print(f"wrote {len(manifest['files'])} fixture files, {total / 1024:.0f} KiB, "
          f"{len(sos_tab)} SOS tables")


if __name__ == "__main__":
    main()
