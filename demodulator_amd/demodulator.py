"""Demodulator: Python face of the C ABI (include/demod_hip.h)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _native
from ._native import DemodError, Params, Span

DEFAULT_BUF_SIZE = 262144  # include/matrix.h:37-39


@dataclass
class Options:
    """The reference's command-line options that reach the hot path, with its
    defaults (/root/reference/src/main.c:102-113)."""

    sample_rate: float = 125000.0   # -S
    lowpass_in: float = 0.0         # -L   (0 = no input lowpass)
    lowpass_out: float = 12500.0    # -l
    epsilon: float = 0.3            # -e x stores x/10
    in_degree: int = 0              # -D
    out_degree: int = 3             # -d
    mode: int = 0x10                # -m | -q<<2 | -c<<4
    buf_size: int = DEFAULT_BUF_SIZE  # DEFAULT_BUF_SIZE shifted by -b

    @classmethod
    def from_cli(cls, S=None, L=None, l=None, D=None, d=None, e=None, m=0, b=0, q=0, c=0):  # noqa: E741
        """Build options the way main()'s getopt loop does (src/main.c:125-183)."""
        o = cls()
        if S is not None:
            o.sample_rate = float(S)
        if L is not None:
            o.lowpass_in = float(L)
        if l is not None:
            o.lowpass_out = float(l)
        if D is not None:
            o.in_degree = int(D)
        if d is not None:
            o.out_degree = int(d)
        if e is not None:
            o.epsilon = float(e) / 10.0
        o.mode |= int(m) | (int(q) << 2) | (int(c) << 4)
        b = int(b)
        if abs(b) < 17:
            o.buf_size = DEFAULT_BUF_SIZE >> -b if b < 1 else DEFAULT_BUF_SIZE << b
        return o

    def c_struct(self) -> Params:
        return Params(self.sample_rate, self.lowpass_in, self.lowpass_out, self.epsilon,
                      self.in_degree, self.out_degree, self.mode, 0, self.buf_size)


def design(options: Options, which: str = "out") -> dict:
    """Host-only filter design (no GPU needed): SOS rows, stencil taps, the
    imaginary-lane bias and the per-chunk head length for an option set.
    which: "out" (-l), "in" (-L, or the -q2 highpass when alone), "pre" (the -q2
    highpass when it runs in front of -L)."""
    lib = _native.load()
    p = options.c_struct()
    rows = np.zeros((16, 6), dtype=np.float32)
    taps = np.zeros(8, dtype=np.float32)
    bias = C.c_float(0)
    n_taps = C.c_int(0)
    head = C.c_int(0)
    n = lib.demod_design_query(C.byref(p), {"out": 0, "in": 1, "pre": 2}[which], rows.ctypes.data, 16,
                               taps.ctypes.data, 8, C.byref(bias), C.byref(n_taps), C.byref(head))
    if n < 0:
        raise DemodError(f"demod_design_query failed ({n}): {lib.demod_last_error(None).decode()}")
    return {"sos": rows[:n].copy(), "taps": taps[:n_taps.value].copy(), "bias_im": float(bias.value),
            "head_outputs": int(head.value), "magnitude_bound": float(lib.demod_design_magnitude_bound(C.byref(p))),
            "unpack_fold_exact": bool(lib.demod_design_unpack_folded(C.byref(p)) == 1), **_stencil_form(lib, p)}


STENCIL_FORMS = {"auto": 0, "plain": 1, "folded": 2, "normalized": 3}


def _stencil_form(lib, p) -> dict:
    """Which form of the complex stencil the streaming kernel gives this option set by default, and the two error
    bounds build_plan compared (include/demod_hip.h, demod_design_stencil_form)."""
    fn = getattr(lib, "demod_design_stencil_form", None)
    if fn is None:      # absent from older builds selected with DEMOD_HIP_LIBRARY
        return {}
    en, ep = C.c_double(0), C.c_double(0)
    form = fn(C.byref(p), C.byref(en), C.byref(ep))
    out = {"stencil_form": {1: "plain", 3: "normalized"}.get(form, "plain"), "normalized_error": float(en.value),
           "plain_error": float(ep.value)}
    fn = getattr(lib, "demod_design_normalized_constants", None)
    if fn is not None:
        a, kr, ki = C.c_float(0), C.c_float(0), C.c_float(0)
        rho = np.zeros(8, dtype=np.float32)
        n = fn(C.byref(p), C.byref(a), C.byref(kr), C.byref(ki), rho.ctypes.data, 8)
        out["normalized"] = {"alpha": float(a.value), "k_re": float(kr.value), "k_im": float(ki.value), "rho": rho[:max(n, 0)].copy()}
    return out


class Demodulator:
    """One demodulation context on one GPU.

    process()       device tensor in -> device tensor out (the hot path)
    process_host()  numpy in -> numpy out through the three-stream host pipeline
    """

    def __init__(self, options: Options | None = None, device: int = 0, **kw):
        self._lib = _native.load()
        self.options = options if options is not None else Options(**kw)
        self.device = int(device)
        self._ctx = C.c_void_p()
        p = self.options.c_struct()
        rc = self._lib.demod_create(C.byref(p), self.device, C.byref(self._ctx))
        if rc != 0:
            msg = self._lib.demod_last_error(None).decode()
            self._ctx = C.c_void_p()
            raise DemodError(f"demod_create failed ({rc}): {msg}")

    # -- lifetime -------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.demod_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            code = getattr(self._lib, "demod_last_error_code", None)
            assert code is None or code(self._ctx) == rc   # the message belongs to this failure
            raise DemodError(f"{what} failed ({rc}): {self._lib.demod_last_error(self._ctx).decode()}")

    # -- introspection ----------------------------------------------------------
    def sos(self, which: str = "out") -> np.ndarray:
        w = 1 if which == "in" else 0
        n = self._lib.demod_get_sos(self._ctx, w, None, 0)
        rows = np.zeros((max(n, 0), 6), dtype=np.float32)
        if n > 0:
            self._lib.demod_get_sos(self._ctx, w, rows.ctypes.data, n)
        return rows

    def taps(self, which: str = "out"):
        w = 1 if which == "in" else 0
        buf = np.zeros(8, dtype=np.float32)
        bias = C.c_float(0)
        n = self._lib.demod_get_taps(self._ctx, w, buf.ctypes.data, 8, C.byref(bias))
        return buf[:max(n, 0)].copy(), float(bias.value)

    @property
    def head_outputs(self) -> int:
        return self._lib.demod_get_head_outputs(self._ctx)

    def set_seamless(self, on: bool = True) -> None:
        """Filter the stream as one chunk (look-ahead continues across chunk
        boundaries); chunk_offset then counts from the start of the stream."""
        self._check(self._lib.demod_set_seamless(self._ctx, 1 if on else 0), "demod_set_seamless")

    def set_output_format(self, fmt: str = "f32", scale: float = 1.0) -> None:
        """"f32" (the reference's output) or "s16": clamp(y * scale, -1, 1) * 32767
        rounded to nearest, the float -> signed 16-bit hop of the reference's README
        pipes fused into the kernel's store (demod_set_output_format)."""
        if fmt not in ("f32", "s16"):
            raise DemodError("output format must be 'f32' or 's16'")
        self._check(self._lib.demod_set_output_format(self._ctx, 1 if fmt == "s16" else 0, scale),
                    "demod_set_output_format")

    @property
    def output_dtype(self):
        return np.int16 if self._lib.demod_output_bytes(self._ctx) == 2 else np.float32

    def set_unpack_fold(self, on: bool = True) -> bool:
        """Ask for the folded form of the unpack (faster, less exact at ill-conditioned samples: not the default).
        True when the design has it (its constant is exactly a float)."""
        rc = self._lib.demod_set_unpack_fold(self._ctx, 1 if on else 0)
        if rc < 0:
            self._check(rc, "demod_set_unpack_fold")
        return rc == 0 and on

    def set_stencil_form(self, form: str = "auto") -> bool:
        """Form of the complex stencil behind the default unpack: "auto" (normalized where the design allows it,
        plain otherwise), "plain", "normalized", "folded" (include/demod_hip.h).  True when the context now runs
        what was asked for."""
        fn = getattr(self._lib, "demod_set_stencil_form", None)
        if fn is None:      # older builds selected with DEMOD_HIP_LIBRARY: plain, or folded through the old call
            return form in ("auto", "plain")
        rc = fn(self._ctx, STENCIL_FORMS[form])
        if rc < 0:
            self._check(rc, "demod_set_stencil_form")
        return rc == 0

    @property
    def stencil_form(self) -> str:
        fn = getattr(self._lib, "demod_stencil_form", None)
        return {1: "plain", 2: "folded", 3: "normalized"}.get(fn(self._ctx), "plain") if fn is not None else "plain"

    def set_head_mode(self, mode: int) -> None:
        """Where chunk heads are replayed: 1 inside the streaming kernel, 0 head kernels, 2 thread-per-chunk head
        kernels only, -1 automatic."""
        fn = getattr(self._lib, "demod_set_head_mode", None)
        if fn is not None:      # absent from older builds selected with DEMOD_HIP_LIBRARY
            self._check(fn(self._ctx, mode), "demod_set_head_mode")

    def set_piece_rows(self, rows: int) -> None:
        self._check(self._lib.demod_set_piece_rows(self._ctx, rows), "demod_set_piece_rows")

    def set_segment_rows(self, rows: int) -> None:
        self._check(self._lib.demod_set_segment_rows(self._ctx, rows), "demod_set_segment_rows")

    # -- the hot path -----------------------------------------------------------
    def process_ptr(self, d_in: int, n_bytes: int, d_out: int, stream: int = 0, *,
                    chunk_offset: int = 0, halo_ptr: int = 0, end_of_stream: bool = True) -> None:
        """Raw-pointer form of demod_process_device (asynchronous on `stream`)."""
        span = Span(chunk_offset, halo_ptr or None, 1 if end_of_stream else 0, 0)
        rc = self._lib.demod_process_device(self._ctx, d_in, n_bytes, d_out, C.byref(span),
                                            stream or None)
        self._check(rc, "demod_process_device")

    def process(self, iq, out=None, *, chunk_offset: int = 0, halo=None, end_of_stream: bool = True):
        """iq: 1-D torch.uint8 CUDA tensor of interleaved I,Q bytes.  Returns a
        torch.float32 (int16 after set_output_format("s16")) CUDA tensor with
        iq.numel() // 4 results.  Runs on torch's current stream."""
        import torch

        odt = torch.int16 if self.output_dtype == np.int16 else torch.float32
        if not (iq.is_cuda and iq.dtype == torch.uint8 and iq.is_contiguous()):
            raise DemodError("process() needs a contiguous CUDA uint8 tensor")
        n = iq.numel()
        if out is None:
            out = torch.empty(n // 4, dtype=odt, device=iq.device)
        elif not (out.is_cuda and out.dtype == odt and out.is_contiguous()
                  and out.numel() >= n // 4):
            raise DemodError(f"out must be a contiguous CUDA {odt} tensor of numel >= n // 4")
        halo_ptr = 0
        if halo is not None:
            if not (halo.is_cuda and halo.dtype == torch.uint8 and halo.numel() >= _native.HALO_BYTES):
                raise DemodError(f"halo must be a CUDA uint8 tensor of >= {_native.HALO_BYTES} bytes")
            halo_ptr = halo.data_ptr()
        stream = torch.cuda.current_stream(iq.device).cuda_stream
        self.process_ptr(iq.data_ptr(), n - (n & 3), out.data_ptr(), stream,
                         chunk_offset=chunk_offset, halo_ptr=halo_ptr, end_of_stream=end_of_stream)
        return out

    def process_host(self, data: np.ndarray, *, chunk_offset: int = 0, halo: np.ndarray | None = None,
                     end_of_stream: bool = True) -> np.ndarray:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n = data.size - (data.size & 3)
        out = np.empty(n // 4, dtype=self.output_dtype)
        hp = None
        if halo is not None:
            halo = np.ascontiguousarray(halo, dtype=np.uint8)
            if halo.size < _native.HALO_BYTES:
                halo = np.concatenate([halo, np.zeros(_native.HALO_BYTES - halo.size, np.uint8)])
            hp = halo.ctypes.data
        span = Span(chunk_offset, hp, 1 if end_of_stream else 0, 0)
        rc = self._lib.demod_process_host(self._ctx, data.ctypes.data, n, out.ctypes.data, C.byref(span))
        self._check(rc, "demod_process_host")
        return out

    # -- pipelined host path (what the backend adapter drives) -------------------
    @property
    def stream_capacity(self) -> int:
        return int(self._lib.demod_stream_capacity())

    def stream_submit(self, data: np.ndarray, *, chunk_offset: int = 0, halo: np.ndarray | None = None,
                      end_of_stream: bool = True) -> None:
        """demod_stream_submit: copy `data` into a pinned staging slot and enqueue upload, hot
        path and download; returns at once.  At most stream_depth pieces may be uncollected."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        hp = None
        if halo is not None:
            halo = np.ascontiguousarray(halo, dtype=np.uint8)
            if halo.size < _native.HALO_BYTES:
                halo = np.concatenate([halo, np.zeros(_native.HALO_BYTES - halo.size, np.uint8)])
            hp = halo.ctypes.data
        span = Span(chunk_offset, hp, 1 if end_of_stream else 0, 0)
        self._check(self._lib.demod_stream_submit(self._ctx, data.ctypes.data, data.size & ~3, C.byref(span)),
                    "demod_stream_submit")

    def stream_collect(self) -> np.ndarray | None:
        """demod_stream_collect: results of the OLDEST submitted piece (a copy), None when nothing is pending."""
        out = C.c_void_p()
        n = C.c_uint64(0)
        self._check(self._lib.demod_stream_collect(self._ctx, C.byref(out), C.byref(n)), "demod_stream_collect")
        if not n.value:
            return None
        dt = self.output_dtype
        buf = (C.c_char * (n.value * np.dtype(dt).itemsize)).from_address(out.value)
        return np.frombuffer(buf, dtype=dt).copy()

    def ring(self, slots: int | None = None) -> "Ring":
        """demod_ring_create with an arena of the ring's own."""
        return Ring(self, slots)

    # -- helpers ----------------------------------------------------------------
    def synth_fill(self, dst, first_byte: int, seed: int) -> None:
        """Fill a CUDA uint8 tensor with the counter-based synthetic IQ stream."""
        import torch

        stream = torch.cuda.current_stream(dst.device).cuda_stream
        rc = self._lib.demod_synth_fill(self._ctx, dst.data_ptr(), first_byte, dst.numel(), seed,
                                        stream or None)
        self._check(rc, "demod_synth_fill")

    def synchronize(self) -> None:
        self._check(self._lib.demod_synchronize(self._ctx), "demod_synchronize")

    def enable_timing(self, on: bool = True) -> None:
        self._lib.demod_enable_timing(self._ctx, 1 if on else 0)

    @property
    def last_call_launches(self) -> int:
        """Dispatches of the fused kernel in the most recent process call."""
        return self._lib.demod_last_call_launches(self._ctx)

    def kernel_ms(self):
        """(average fused-kernel milliseconds per process call, calls) since the last call."""
        n = C.c_int(0)
        ms = self._lib.demod_kernel_ms(self._ctx, C.byref(n))
        return float(ms), int(n.value)

    def selftest_atan2(self, y: np.ndarray, x: np.ndarray) -> np.ndarray:
        y = np.ascontiguousarray(y, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(y)
        self._check(self._lib.demod_selftest_atan2(self._ctx, y.ctypes.data, x.ctypes.data,
                                                   out.ctypes.data, y.size), "demod_selftest_atan2")
        return out

    def selftest_q1_offsets(self, n_pairs: int):
        """-q1: (offsets of the last call as an (n_pairs, 2) float32 array, the state it left as a (2,) array)."""
        pairs = np.zeros((n_pairs, 2), dtype=np.float32)
        state = np.zeros(2, dtype=np.float32)
        self._check(self._lib.demod_selftest_q1_offsets(self._ctx, pairs.ctypes.data if n_pairs else None, n_pairs, state.ctypes.data),
                    "demod_selftest_q1_offsets")
        return pairs, state

    def selftest_lane_shift(self, v: np.ndarray, halo: float) -> np.ndarray:
        v = np.ascontiguousarray(v, dtype=np.float32)
        out = np.empty(64, dtype=np.float32)
        self._check(self._lib.demod_selftest_lane_shift(self._ctx, v.ctypes.data, halo,
                                                        out.ctypes.data), "demod_selftest_lane_shift")
        return out


class Ring:
    """demod_ring_*: the producer/consumer form of the host path (what the backend adapter
    drives).  put() = demod_ring_input + copy + demod_ring_commit from the producer's thread;
    take() = demod_ring_collect + copy + demod_ring_release from the consumer's thread."""

    def __init__(self, dem: Demodulator, slots: int | None = None):
        self._d = dem
        self._lib = dem._lib
        self._ring = C.c_void_p()
        self.chunk = dem.options.buf_size
        arena = None
        n = 0
        if slots is not None:   # a caller-owned pinned arena, as allocateBuffer makes one
            n = int(slots)
            arena = C.c_void_p()
            rc = self._lib.demod_alloc_pinned(C.byref(arena), n * self.chunk)
            if rc != 0:
                raise DemodError(f"demod_alloc_pinned failed ({rc})")
        self._arena = arena
        dem._check(self._lib.demod_ring_create(dem._ctx, arena, n, C.byref(self._ring)), "demod_ring_create")

    def put(self, data: np.ndarray, final: bool = False) -> None:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n = data.size & ~3
        if n:
            dst = self._lib.demod_ring_input(self._ring)
            C.memmove(dst, data.ctypes.data, n)
        self._d._check(self._lib.demod_ring_commit(self._ring, n, 1 if final else 0), "demod_ring_commit")

    def take(self) -> np.ndarray | None:
        out = C.c_void_p()
        n = C.c_uint64(0)
        self._d._check(self._lib.demod_ring_collect(self._ring, C.byref(out), C.byref(n)), "demod_ring_collect")
        if not n.value:
            return None
        dt = self._d.output_dtype
        buf = (C.c_char * (n.value * np.dtype(dt).itemsize)).from_address(out.value)
        res = np.frombuffer(buf, dtype=dt).copy()
        self._d._check(self._lib.demod_ring_release(self._ring), "demod_ring_release")
        return res

    def close(self) -> None:
        if self._ring.value:
            self._lib.demod_ring_destroy(self._ring)
            self._ring = C.c_void_p()
        if self._arena is not None and self._arena.value:
            self._lib.demod_free_pinned(self._arena)
            self._arena = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def plan_shards_c(total_bytes: int, n: int, chunk_bytes: int, phase_bytes: int = 0, align: int = 0):
    """demod_plan_shards (the C planner; demodulator_amd.sharding.plan_shards is its Python twin)."""
    lib = _native.load()
    arr = (_native.Shard * n)()
    rc = lib.demod_plan_shards(total_bytes, n, chunk_bytes, phase_bytes, align, arr)
    if rc != 0:
        raise DemodError(f"demod_plan_shards failed ({rc})")
    return list(arr)


class MultiDemodulator:
    """demod_multi_*: one process, several GPUs (one context and stream per device)."""

    def __init__(self, options: Options, devices):
        self._lib = _native.load()
        self.options = options
        self.devices = [int(d) for d in devices]
        self._m = C.c_void_p()
        p = options.c_struct()
        devs = (C.c_int * len(self.devices))(*self.devices)
        rc = self._lib.demod_multi_create(C.byref(p), devs, len(self.devices), C.byref(self._m))
        if rc != 0:
            raise DemodError(f"demod_multi_create failed ({rc}): {self._lib.demod_last_error(None).decode()}")

    def plan(self, total_bytes: int, phase_bytes: int = 0):
        arr = (_native.Shard * len(self.devices))()
        rc = self._lib.demod_multi_plan(self._m, total_bytes, phase_bytes, arr)
        if rc != 0:
            raise DemodError(f"demod_multi_plan failed ({rc})")
        return arr

    def plan_channels(self, n_channels: int, bytes_per_channel: int):
        arr = (_native.Shard * len(self.devices))()
        rc = self._lib.demod_plan_channels(n_channels, bytes_per_channel, len(self.devices), self.options.buf_size, arr)
        if rc != 0:
            raise DemodError(f"demod_plan_channels failed ({rc})")
        return arr

    def process(self, ins, outs, shards) -> None:
        """ins / outs: one CUDA tensor per device (shard i on device i).  Asynchronous; synchronize() waits."""
        n = len(self.devices)
        a = (C.c_void_p * n)(*[t.data_ptr() for t in ins])
        b = (C.c_void_p * n)(*[t.data_ptr() for t in outs])
        rc = self._lib.demod_multi_process_device(self._m, a, b, shards)
        if rc != 0:
            raise DemodError(f"demod_multi_process_device failed ({rc}): {self._lib.demod_multi_last_error(self._m).decode()}")

    def synchronize(self) -> None:
        if self._lib.demod_multi_synchronize(self._m) != 0:
            raise DemodError(self._lib.demod_multi_last_error(self._m).decode())

    @property
    def halo_transport(self) -> str:
        return self._lib.demod_multi_halo_transport(self._m).decode()

    def enable_timing(self, on: bool = True) -> None:
        """HIP-event timing of the streaming kernel on every device's context (demod_enable_timing)."""
        for i in range(len(self.devices)):
            self._lib.demod_enable_timing(self._lib.demod_multi_context(self._m, i), 1 if on else 0)

    def kernel_ms(self, i: int):
        """(average fused-kernel milliseconds per call, calls, dispatches of the last call) of device i's context."""
        ctx = self._lib.demod_multi_context(self._m, i)
        n = C.c_int(0)
        ms = self._lib.demod_kernel_ms(ctx, C.byref(n))
        return float(ms), int(n.value), int(self._lib.demod_last_call_launches(ctx))

    def close(self) -> None:
        if self._m.value:
            self._lib.demod_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
