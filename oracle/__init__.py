"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end to the CPU restatement of the reference hot path
(oracle/demod_oracle.c) and, where it has been built, to the real reference
compiled from /root/reference (oracle/_ref/libdemod_ref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  demodulator_amd/ never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "libdemod_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libdemod_ref.so")
_REF_FMA_SO = os.path.join(_HERE, "_ref", "libdemod_ref_fma.so")


class _Params(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_longdouble),
        ("lowpass_in", C.c_longdouble),
        ("lowpass_out", C.c_longdouble),
        ("epsilon", C.c_longdouble),
        ("in_degree", C.c_size_t),
        ("out_degree", C.c_size_t),
        ("mode", C.c_uint),
        ("buf_size", C.c_size_t),
    ]


@dataclass
class Flags:
    """Command-line options of the reference that reach processMatrix
    (src/main.c:102-113,125-183)."""

    sample_rate: float = 125000.0   # -S
    lowpass_in: float = 0.0         # -L
    lowpass_out: float = 12500.0    # -l
    epsilon: float = 0.3            # -e x  -> x/10
    in_degree: int = 0              # -D
    out_degree: int = 3             # -d
    mode: int = 0x10                # -m / -c / -q bits
    buf_size: int = 262144          # DEFAULT_BUF_SIZE << -b

    def c_struct(self) -> _Params:
        return _Params(self.sample_rate, self.lowpass_in, self.lowpass_out, self.epsilon,
                       self.in_degree, self.out_degree, self.mode, self.buf_size)


def build(force: bool = False) -> None:
    """Compile the restatement (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_ORACLE_SO) or (
            os.path.getmtime(_ORACLE_SO) < os.path.getmtime(os.path.join(_HERE, "demod_oracle.c"))):
        subprocess.run(["make", "-C", _HERE, "libdemod_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/src/matrix.c") and (force or not os.path.exists(_REF_SO) or not os.path.exists(_REF_FMA_SO)):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, stdout=subprocess.DEVNULL)


class _Lib:
    """One loaded implementation: prefix 'orc' (restatement) or 'ref' (reference)."""

    def __init__(self, path: str, prefix: str):
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path)
        f = self._fn
        f("design_sos", None, [C.c_uint, C.c_size_t, C.c_longdouble, C.c_longdouble,
                               C.c_longdouble, C.c_void_p])
        f("unpack_shift", None, [C.c_void_p, C.c_size_t, C.c_void_p])
        f("unpack_normalize", None, [C.c_void_p, C.c_size_t, C.c_void_p])
        f("discriminate", None, [C.c_void_p, C.c_size_t, C.c_void_p])
        if prefix == "orc":
            f("unpack_correct_iq", None, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_float, C.c_void_p])
        f("filter_real", None, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p])
        f("filter_complex", None, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p])
        f("process_stream", C.c_longlong, [C.POINTER(_Params), C.c_void_p, C.c_size_t,
                                           C.c_void_p])
        if prefix == "orc":
            f("process_stream_mt", C.c_longlong, [C.POINTER(_Params), C.c_void_p, C.c_size_t,
                                                  C.c_void_p, C.c_int])
            f("synth_bytes", None, [C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p])

    def _fn(self, name, restype, argtypes):
        fn = getattr(self.lib, f"{self.prefix}_{name}")
        fn.restype = restype
        fn.argtypes = argtypes
        setattr(self, "_" + name, fn)

    # -- setup -------------------------------------------------------------
    def design_sos(self, kind: int, degree: int, fc: float, fs: float,
                   epsilon: float = 0.3) -> np.ndarray:
        rows = (degree >> 1) + (degree & 1)
        sos = np.zeros((rows, 6), dtype=np.float32)
        self._design_sos(kind, degree, fc, fs, epsilon, sos.ctypes.data)
        return sos

    # -- primitives ----------------------------------------------------------
    def unpack_shift(self, data: np.ndarray) -> np.ndarray:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.zeros(data.size, dtype=np.float32)
        self._unpack_shift(data.ctypes.data, data.size, out.ctypes.data)
        return out

    def unpack_normalize(self, data: np.ndarray) -> np.ndarray:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.zeros(data.size, dtype=np.float32)
        self._unpack_normalize(data.ctypes.data, data.size, out.ctypes.data)
        return out

    def unpack_correct_iq(self, data: np.ndarray, chunk_bytes: int, sample_rate: float) -> np.ndarray:
        """-q1 (correctIq, src/matrix.c:120-140) alone: the corrected samples byte - running offset of a whole stream,
        chunk after chunk, the offsets starting at zero and carried from chunk to chunk as the reference's static
        pair is.  (The restatement's function: bit-identical to the reference's, tests/test_oracle_golden.py.)"""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.zeros(data.size, dtype=np.float32)
        off = np.zeros(2, dtype=np.float32)
        esr = np.float32(50.0 / sample_rate)
        for pos in range(0, data.size & ~3, chunk_bytes):
            ln = min(chunk_bytes, (data.size & ~3) - pos)
            self._unpack_correct_iq(data.ctypes.data + pos, ln, out.ctypes.data + 4 * pos, esr, off.ctypes.data)
        self.last_correct_iq_state = off      # the pair the next chunk would start from
        return out

    def discriminate(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(x.size // 4, dtype=np.float32)
        self._discriminate(x.ctypes.data, x.size, out.ctypes.data)
        return out

    def filter_real(self, x: np.ndarray, sos: np.ndarray, y0: np.ndarray | None = None,
                    pad: int = 64) -> np.ndarray:
        """y (len + pad floats, caller-visible overflow included)."""
        n = x.size
        sos = np.ascontiguousarray(sos, dtype=np.float32)
        xx = np.zeros(n + pad, dtype=np.float32)
        xx[:n] = x
        y = np.zeros(n + pad, dtype=np.float32)
        if y0 is not None:
            y[:y0.size] = y0
        self._filter_real(xx.ctypes.data, y.ctypes.data, n, sos.shape[0], sos.ctypes.data)
        return y

    def filter_complex(self, x: np.ndarray, sos: np.ndarray, pad: int = 64) -> np.ndarray:
        """x: interleaved re/im floats (no aliasing here: x is zero-extended)."""
        n = x.size
        sos = np.ascontiguousarray(sos, dtype=np.float32)
        xx = np.zeros(n + pad, dtype=np.float32)
        xx[:n] = x
        y = np.zeros(n + pad, dtype=np.float32)
        self._filter_complex(xx.ctypes.data, y.ctypes.data, n, sos.shape[0], sos.ctypes.data)
        return y

    # -- the path ------------------------------------------------------------
    def process(self, flags: Flags, data: np.ndarray, threads: int = 1) -> np.ndarray:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.zeros(data.size // 4 + 4, dtype=np.float32)
        p = flags.c_struct()
        if self.prefix == "ref" and ((flags.mode >> 2) & 3) in (1, 2):
            # -q1 / -q2: the reference keeps static state (running offsets; a buffer sized by
            # its first call), so every stream gets a process of its own (oracle/ref_stream_main.c)
            exe = os.path.join(os.path.dirname(self.path), "ref_stream_fma" if self.path.endswith("_fma.so") else "ref_stream")
            msg = bytes(p) + np.uint64(data.size).tobytes() + data.tobytes()
            r = subprocess.run([exe], input=msg, stdout=subprocess.PIPE, check=False)
            if r.returncode != 0:
                raise ValueError(f"option set not supported by the ref oracle: {flags}")
            return np.frombuffer(r.stdout, dtype=np.float32).copy()
        if threads > 1 and self.prefix == "orc":
            n = self._process_stream_mt(C.byref(p), data.ctypes.data, data.size,
                                        out.ctypes.data, threads)
        else:
            n = self._process_stream(C.byref(p), data.ctypes.data, data.size, out.ctypes.data)
        if n < 0:
            raise ValueError(f"option set not supported by the {self.prefix} oracle: {flags}")
        return out[:n]

    def synth_bytes(self, seed: int, first_byte: int, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.uint8)
        self._synth_bytes(seed, first_byte, n, out.ctypes.data)
        return out


_cache: dict[str, _Lib | None] = {}


def restatement() -> _Lib:
    """The CPU restatement (always available; built on demand)."""
    if "orc" not in _cache:
        build()
        _cache["orc"] = _Lib(_ORACLE_SO, "orc")
    return _cache["orc"]


def restatement_fma() -> _Lib | None:
    """The restatement compiled with FMA contraction (what the reference's IS_NATIVE build
    does on an FMA machine), or None when this CPU has no FMA."""
    if "fma" not in _cache:
        path = os.path.join(_HERE, "libdemod_oracle_fma.so")
        try:
            has_fma = " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
        except OSError:
            has_fma = False
        if has_fma and not os.path.exists(path):
            subprocess.run(["make", "-C", _HERE, "libdemod_oracle_fma.so"], check=False, stdout=subprocess.DEVNULL)
        _cache["fma"] = _Lib(path, "orc") if has_fma and os.path.exists(path) else None
    return _cache["fma"]


def _cpu_has_fma() -> bool:
    try:
        return " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


def reference_fma() -> _Lib | None:
    """The real reference compiled as its IS_NATIVE build comes out on an FMA machine
    (oracle/Makefile, _ref/libdemod_ref_fma.so), or None when it has not been built (no
    /root/reference and no prebuilt copy) or this CPU has no FMA."""
    if "ref_fma" not in _cache:
        build()
        _cache["ref_fma"] = _Lib(_REF_FMA_SO, "ref") if _cpu_has_fma() and os.path.exists(_REF_FMA_SO) else None
    return _cache["ref_fma"]


def reference() -> _Lib | None:
    """The real reference compiled from /root/reference, or None when neither
    the sources nor a prebuilt oracle/_ref are present."""
    if "ref" not in _cache:
        build()
        _cache["ref"] = _Lib(_REF_SO, "ref") if os.path.exists(_REF_SO) else None
    return _cache["ref"]


# ---------------------------------------------------------------------------
# synthetic input in numpy (same definition as orc_synth_bytes, vectorised)
# ---------------------------------------------------------------------------
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def synth_bytes(seed: int, first_byte: int, n: int) -> np.ndarray:
    """Bytes [first_byte, first_byte+n) of the counter-based stream `seed`
    (SURVEY.md section 8d): byte i = byte (i & 7) of splitmix64 word (i >> 3)."""
    w0 = first_byte >> 3
    w1 = (first_byte + n + 7) >> 3
    with np.errstate(over="ignore"):
        j = np.arange(w0 + 1, w1 + 1, dtype=np.uint64)
        z = np.uint64(seed) + j * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    b = z.view(np.uint8)  # little-endian host: byte k of word = bits 8k..8k+7
    off = first_byte - (w0 << 3)
    return b[off:off + n].copy()
