"""CPU-side checks of the product's host logic and of the C-ABI library itself
(no compute calls -- those need the GPU): the library loads, exports every
symbol include/demod_hip.h declares, designs the same filters as the reference
and refuses the option sets the reference leaves undefined."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import demodulator_amd as dm
from demodulator_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "demod_hip.h")).read()
    declared = set(re.findall(r"\b(demod_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"demod_params", "demod_span", "demod_ctx"}
    assert declared, "no declarations parsed"
    lib = C.CDLL(_native.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in demod_hip.h but not exported"
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)


def test_backend_library_exports_reference_boundary():
    path = os.path.join(ROOT, "demodulator_amd", "libdemod_backend.so")
    if not os.path.exists(path):
        pytest.skip("libdemod_backend.so not built")
    lib = C.CDLL(path)
    # the two symbols of the reference's backend boundary, include/matrix.h:59-62
    assert hasattr(lib, "processMatrix") and hasattr(lib, "allocateBuffer")


def test_abi_struct_sizes():
    assert C.sizeof(_native.Params) == 56
    assert C.sizeof(_native.Span) == 24
    assert _native.load().demod_abi_version() == _native.ABI_VERSION


def test_design_matches_reference_sos(manifest):
    """Product filter design (plan.cpp) == reference's processFilterOption, hex-exact,
    for the lowpass kinds the path uses (mode bit0/bit1)."""
    for row in manifest["sos"]:
        if row["kind"] not in (0, 1):
            continue
        deg = row["degree"]
        # output filter, kind selected by mode bit 0
        opt = dm.Options(sample_rate=row["fs"], lowpass_out=row["fc"], out_degree=deg,
                         epsilon=row["epsilon"], mode=0x10 | row["kind"], buf_size=4096)
        got = dm.design(opt, "out")["sos"]
        want = np.array([[float.fromhex(v) for v in r] for r in row["sos_hex"]], dtype=np.float32)
        assert np.array_equal(bits(got), bits(want)), row
        # input filter, kind selected by mode bit 1
        opt = dm.Options(sample_rate=row["fs"], lowpass_in=row["fc"], in_degree=deg,
                         epsilon=row["epsilon"], mode=0x10 | (row["kind"] << 1), buf_size=4096)
        got = dm.design(opt, "in")["sos"]
        assert np.array_equal(bits(got), bits(want)), row


def test_taps_and_bias_survey_values():
    # SURVEY.md Appendix A.3 table
    d = dm.design(dm.Options())
    assert np.allclose(d["taps"], [1.760042, -0.5457233], rtol=1e-7)
    assert d["head_outputs"] == 3
    o = dm.Options(lowpass_in=12500.0, lowpass_out=6500.0)
    assert np.allclose(dm.design(o)["taps"], [2.3492503, -0.72343695], rtol=1e-7)
    din = dm.design(o, "in")
    assert np.allclose(din["taps"], [1.760042, -0.5457233], rtol=1e-7)
    assert din["bias_im"] == -2.0 and din["head_outputs"] == 6
    # imaginary-lane bias -1/-2/-5/-10 for 1..4 sections; 1, 2, 3 taps
    for deg, bias, taps in ((1, -1, 1), (2, -1, 2), (3, -2, 2), (5, -5, 2), (6, -5, 3), (7, -10, 3),
                            (8, -10, 3)):
        di = dm.design(dm.Options(lowpass_in=12500.0, in_degree=deg), "in")
        assert di["bias_im"] == bias and di["taps"].size == taps, (deg, di)


def test_taps_reproduce_the_recurrence_in_the_interior():
    """The stencil derived in plan.cpp, applied with numpy, equals the oracle's
    literal recurrence away from the chunk head (host logic, not the GPU path)."""
    import oracle
    orc = oracle.restatement()
    rng = np.random.default_rng(3)
    for deg in range(1, 9):
        d = dm.design(dm.Options(out_degree=deg, lowpass_out=6500.0))
        x = rng.uniform(-3.14, 3.14, 4096).astype(np.float32)
        y = orc.filter_real(x, d["sos"])[:4096]
        xp = np.concatenate([x, np.zeros(8, np.float32)]).astype(np.float64)
        model = sum(float(h) * xp[j:j + 4096] for j, h in enumerate(d["taps"]))
        H = d["head_outputs"]
        assert np.max(np.abs(model[H:] - y[H:])) < 2e-5 * max(1.0, np.abs(d["taps"]).sum()), deg


def test_rejected_option_sets():
    bad = [
        dm.Options(in_degree=3),                         # -D without -L (src/matrix.c:227-231)
        dm.Options(mode=0x14, sample_rate=90.0),         # -q1 with 100/S >= 1: the offset recurrence does not contract
        dm.Options(mode=0x18, lowpass_in=12500.0, in_degree=14),  # -q2 + -L with a long input stencil
        dm.Options(mode=0x00),                           # demod mode 0
        dm.Options(out_degree=0),
        dm.Options(out_degree=17),
        dm.Options(buf_size=3000),
    ]
    for o in bad:
        with pytest.raises(dm.DemodError):
            dm.design(o)


def test_options_from_cli_matches_getopt_rules():
    o = dm.Options.from_cli(S=96000, l=6500, L=12500, b=-8, m=3)
    assert (o.sample_rate, o.lowpass_out, o.lowpass_in) == (96000.0, 6500.0, 12500.0)
    assert o.buf_size == 1024 and o.mode == 0x13
    assert dm.Options.from_cli(b=2).buf_size == 262144 << 2
    assert dm.Options.from_cli(b=17).buf_size == 262144          # |b| >= 17 ignored (src/main.c:166)
    assert dm.Options.from_cli(e=3).epsilon == pytest.approx(0.3)
    assert dm.Options.from_cli(q=3).mode == 0x1C


def test_context_creation_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dm.DemodError, match="no HIP device"):
        dm.Demodulator()


def test_cli_fails_loudly_without_gpu():
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = os.path.join(ROOT, "demodulator_amd", "demod")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    r = subprocess.run([exe, "-i", "-", "-o", "-"], input=b"\x80" * 4096, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=60)
    assert r.returncode != 0 and b"no ROCm-capable device" in r.stderr   # no silent CPU path
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode != 0                       # argc < 3, src/main.c:122-124


def test_hand_counted_waits_cover_every_asm_load():
    """Static replay of the generated gfx950 assembly of every demod_rows
    instantiation: no instruction touches a register of an inline-asm row load
    before the hand-counted s_waitcnt that covers it (tools/check_asm_waits.py)."""
    import shutil
    import subprocess
    import sys
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_waits.py")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()
    assert b" 0 violations" in r.stdout


def test_folded_unpack_is_offered_only_where_its_constant_is_exact():
    """The folded form of the unpack (an option, not the default: DESIGN.md section 5) is available
    only for designs whose constant -128 * sum(h) (and the imaginary lane's, bias included) is
    exactly a float: the BASELINE designs are, -L 9600 at 125 kS/s is half an ulp (7.6e-6) off."""
    from fractions import Fraction

    def exact(opt):
        d = dm.design(opt, "in")
        s = sum(Fraction(float(np.float32(t))) for t in d["taps"])
        re, im = -128 * s, Fraction(float(np.float32(d["bias_im"]))) - 128 * s
        return all(Fraction(float(np.float32(float(c)))) == c for c in (re, im))

    L = dict(lowpass_in=12500.0, lowpass_out=6500.0)
    assert dm.design(dm.Options(**L))["unpack_fold_exact"] and exact(dm.Options(**L))                       # config 3 / the target
    assert not dm.design(dm.Options(lowpass_in=9600.0))["unpack_fold_exact"] and not exact(dm.Options(lowpass_in=9600.0))
    assert not dm.design(dm.Options())["unpack_fold_exact"]                                                  # no complex stage: nothing to fold into
    assert not dm.design(dm.Options(mode=0x18, **L))["unpack_fold_exact"]                                   # -q2 never folds
    rng = np.random.default_rng(3)
    n_exact = 0
    for _ in range(200):
        opt = dm.Options(sample_rate=float(rng.choice([96000.0, 125000.0, 192000.0, 250000.0])), lowpass_in=float(rng.integers(3000, 20000)),
                         in_degree=int(rng.integers(1, 9)), mode=0x10 | int(rng.integers(0, 4)))
        try:
            d = dm.design(opt)
        except dm.DemodError:
            continue
        assert d["unpack_fold_exact"] == exact(opt), opt
        n_exact += d["unpack_fold_exact"]
    assert 10 < n_exact < 190


def test_normalized_stencil_constants_are_exact_where_the_form_is_offered():
    """The streaming kernel's default for -q0 with -L (round 3): the complex stencil divided by its first tap, its
    constant riding on the unpack X = alpha * (sample + 128) - (k_re, k_im) (plan.cpp, price_normalised_stencil).
    What the kernel relies on, checked in exact rational arithmetic for every design the form is offered for:
      * alpha has 16 significant bits and k_re = 128 alpha exactly: the real lane is alpha * x to the bit;
      * alpha * U - k_im is a float for EVERY byte value U (no rounding in the unpack), and below 2^8;
      * the bias this applies, h0 (1 + sum rho) (128 - k_im / alpha), is the design's to within half an ulp of it;
      * the taps h0 rho_j are the design's to within what build_plan reports (normalized_error), which is at most an
        eighth of what the plain form's intermediate rounding can do (plain_error);
    and the form is never offered for -q2 or without a complex stage.  BASELINE config 3 / the target has it."""
    from fractions import Fraction as F

    def f32(x):
        return F(float(np.float32(float(x))))

    def check(opt):
        d = dm.design(opt)
        din = dm.design(opt, "in")
        c = d["normalized"]
        alpha, k_re, k_im = F(c["alpha"]), F(c["k_re"]), F(c["k_im"])
        rho = [F(float(r)) for r in c["rho"]]
        h = [F(float(np.float32(t))) for t in din["taps"]]
        bias = F(float(np.float32(din["bias_im"])))
        assert (alpha * 2 ** 15).denominator == 1 and 1 <= alpha < F(3, 2)
        assert k_re == 128 * alpha and 128 <= k_im < 256
        for U in range(256):
            x = alpha * U - k_im
            assert f32(x) == x and abs(x) < 256, (opt, U)
            assert f32(alpha * U - k_re) == alpha * (U - 128)
        assert rho[0] == 1 and len(rho) == len(h)
        applied = h[0] * sum(rho) * (128 - k_im / alpha)
        ulp = F(2) ** (int(np.floor(np.log2(abs(float(bias))))) - 23)
        assert abs(applied - bias) <= ulp / 2, (opt, float(applied - bias))
        taps_err = sum(abs(h[0] * r - hj) for r, hj in zip(rho, h)) * 128
        # (normalized_error as reported includes the bias term bounded above; offered at an eighth of the plain form's price)
        assert float(taps_err) <= d["normalized_error"] + 1e-12 and 8 * d["normalized_error"] <= d["plain_error"] + 8 * float(ulp / 2), opt

    L = dict(lowpass_in=12500.0, lowpass_out=6500.0)
    assert dm.design(dm.Options(**L))["stencil_form"] == "normalized"
    check(dm.Options(**L))
    assert dm.design(dm.Options())["stencil_form"] == "plain"                    # no complex stage
    assert dm.design(dm.Options(mode=0x18, **L))["stencil_form"] == "plain"      # -q2: highpass taps, plain
    assert dm.design(dm.Options(mode=0x1C, **L))["stencil_form"] == "plain"      # -q3: another unpack
    rng = np.random.default_rng(8)
    offered = 0
    for _ in range(150):
        opt = dm.Options(sample_rate=float(rng.choice([96000.0, 125000.0, 192000.0, 250000.0])), lowpass_in=float(rng.integers(3000, 30000)),
                         in_degree=int(rng.integers(1, 17)), mode=0x10 | int(rng.integers(0, 4)))
        try:
            d = dm.design(opt)
        except dm.DemodError:
            continue
        if d["stencil_form"] == "normalized":
            offered += 1
            check(opt)
    assert offered > 8      # (about one design in six has quotients that close to floats)


def test_every_accepted_design_is_proven_finite():
    """The reference maps a NaN discriminator value to 0 (/root/reference/src/matrix.c:174).
    NaN needs non-finite arithmetic; build_plan bounds every intermediate of the reference's
    recurrences from the design's coefficients (plan_magnitude_bound) and refuses what could
    reach 1e30, and designs whose coefficients are themselves non-finite.  So: whenever the
    oracle's output is non-finite the design is refused, every accepted design's outputs are
    finite and below the proven bound -- the NaN rule never fires, here or in the reference."""
    import oracle
    orc = oracle.restatement()
    rng = np.random.default_rng(3)
    accepted = refused = 0
    for it in range(240):
        wide = it % 2 == 0          # half the cases far outside sensible audio designs
        opt = dm.Options(sample_rate=float(10 ** rng.uniform(3, 8)) if wide else float(rng.choice([96000., 125000., 192000.])),
                         lowpass_in=(float(10 ** rng.uniform(0, 6)) if wide else 12500.0) if rng.random() < .5 else 0.0,
                         lowpass_out=float(10 ** rng.uniform(0, 6)) if wide else float(rng.choice([6500., 12500.])),
                         in_degree=0, out_degree=int(rng.integers(1, 17)),
                         mode=0x10 | int(rng.integers(0, 4)) | int(rng.choice([0, 0, 8, 12])),
                         epsilon=float(10 ** rng.uniform(-4, 1.5)) if wide else 0.3, buf_size=4096)
        data = rng.integers(0, 256, 2 * opt.buf_size, dtype=np.uint8)
        data[:1024] = rng.choice([0, 255, 127, 128], 1024)      # extremes as well
        try:
            want = orc.process(oracle.Flags(**opt.__dict__), data)
        except ValueError:
            continue
        try:
            bound = dm.design(opt)["magnitude_bound"]
        except dm.DemodError as e:
            refused += 1
            assert "non-finite" in str(e) or "overflow" in str(e) or "-q2 together with -L" in str(e), e
            continue
        accepted += 1
        assert np.all(np.isfinite(want)), opt
        assert float(np.abs(want).max()) <= bound < 1e30, (opt, np.abs(want).max(), bound)
    assert accepted >= 100 and refused >= 10
