"""ctypes binding of the C ABI in include/demod_hip.h (libdemod_hip.so).

The library is the product: there is no Python or CPU implementation of the
path behind it.  If it is missing, or no MI355X is visible, loading or context
creation raises -- nothing falls back silently.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DEMOD_HIP_LIBRARY selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("DEMOD_HIP_LIBRARY") or os.path.join(_HERE, "libdemod_hip.so")

ABI_VERSION = 4
HALO_BYTES = 64
ROW_BYTES = 1024

# every symbol include/demod_hip.h declares
EXPORTS = (
    "demod_default_params", "demod_abi_version", "demod_create", "demod_destroy",
    "demod_last_error", "demod_process_device", "demod_process_host", "demod_stream_capacity",
    "demod_stream_submit", "demod_stream_collect", "demod_alloc_pinned",
    "demod_free_pinned", "demod_alloc_device", "demod_free_device", "demod_copy_to_device",
    "demod_copy_to_host", "demod_synchronize", "demod_get_sos", "demod_get_taps",
    "demod_get_head_outputs", "demod_design_query", "demod_set_seamless", "demod_set_output_format",
    "demod_output_bytes", "demod_set_segment_rows", "demod_set_piece_rows", "demod_synth_fill",
    "demod_selftest_atan2", "demod_selftest_lane_shift", "demod_selftest_q1_offsets", "demod_kernel_ms", "demod_last_call_launches",
    "demod_enable_timing", "demod_last_error_code", "demod_design_magnitude_bound", "demod_design_unpack_folded", "demod_set_unpack_fold", "demod_set_head_mode",
    "demod_set_stencil_form", "demod_stencil_form", "demod_design_stencil_form", "demod_design_normalized_constants",
    "demod_stream_depth", "demod_ring_slots_for", "demod_ring_create", "demod_ring_destroy", "demod_ring_input",
    "demod_ring_commit", "demod_ring_collect", "demod_ring_release",
    "demod_plan_shards", "demod_plan_channels", "demod_multi_create", "demod_multi_destroy", "demod_multi_devices",
    "demod_multi_context", "demod_multi_stream", "demod_multi_plan", "demod_multi_process_device",
    "demod_multi_synchronize", "demod_multi_halo_transport", "demod_multi_last_error",
)


class Params(C.Structure):
    """struct demod_params"""
    _fields_ = [
        ("sample_rate", C.c_double), ("lowpass_in", C.c_double), ("lowpass_out", C.c_double),
        ("epsilon", C.c_double), ("in_degree", C.c_uint32), ("out_degree", C.c_uint32),
        ("mode", C.c_uint32), ("reserved", C.c_uint32), ("buf_size", C.c_uint64),
    ]


class Span(C.Structure):
    """struct demod_span"""
    _fields_ = [
        ("chunk_offset", C.c_uint64), ("halo", C.c_void_p), ("end_of_stream", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class Shard(C.Structure):
    """struct demod_shard"""
    _fields_ = [
        ("start", C.c_uint64), ("length", C.c_uint64), ("chunk_offset", C.c_uint64), ("needs_halo", C.c_int32),
        ("end_of_stream", C.c_int32), ("halo_src", C.c_int32), ("halo_dst", C.c_int32),
    ]


class DemodError(RuntimeError):
    pass


_lib = None


def load() -> C.CDLL:
    """Load libdemod_hip.so (once).  torch, when installed, is imported first so
    that both share one HIP runtime in the process."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DemodError(
            f"{LIB_PATH} is missing: build it with `make lib` (hipcc --offload-arch=gfx950). "
            "There is no fallback implementation of the demodulation path.")
    try:
        import torch  # noqa: F401  (its bundled libamdhip64 must be the one in the process)
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    sig = {
        "demod_default_params": (None, [C.POINTER(Params)]),
        "demod_abi_version": (i32, []),
        "demod_create": (i32, [C.POINTER(Params), i32, C.POINTER(vp)]),
        "demod_destroy": (None, [vp]),
        "demod_last_error": (C.c_char_p, [vp]),
        "demod_process_device": (i32, [vp, vp, u64, vp, C.POINTER(Span), vp]),
        "demod_process_host": (i32, [vp, vp, u64, vp, C.POINTER(Span)]),
        "demod_stream_capacity": (u64, []),
        "demod_stream_submit": (i32, [vp, vp, u64, C.POINTER(Span)]),
        "demod_stream_collect": (i32, [vp, C.POINTER(vp), C.POINTER(u64)]),
        "demod_alloc_pinned": (i32, [C.POINTER(vp), C.c_size_t]),
        "demod_free_pinned": (None, [vp]),
        "demod_alloc_device": (i32, [vp, C.POINTER(vp), C.c_size_t]),
        "demod_free_device": (None, [vp, vp]),
        "demod_copy_to_device": (i32, [vp, vp, vp, C.c_size_t]),
        "demod_copy_to_host": (i32, [vp, vp, vp, C.c_size_t]),
        "demod_synchronize": (i32, [vp]),
        "demod_get_sos": (i32, [vp, i32, vp, i32]),
        "demod_get_taps": (i32, [vp, i32, vp, i32, C.POINTER(C.c_float)]),
        "demod_get_head_outputs": (i32, [vp]),
        "demod_design_query": (i32, [C.POINTER(Params), i32, vp, i32, vp, i32, C.POINTER(C.c_float),
                                     C.POINTER(i32), C.POINTER(i32)]),
        "demod_set_seamless": (i32, [vp, i32]),
        "demod_set_output_format": (i32, [vp, i32, C.c_float]),
        "demod_output_bytes": (i32, [vp]),
        "demod_set_segment_rows": (i32, [vp, u32]),
        "demod_set_piece_rows": (i32, [vp, u32]),
        "demod_synth_fill": (i32, [vp, vp, u64, u64, u64, vp]),
        "demod_selftest_atan2": (i32, [vp, vp, vp, vp, u32]),
        "demod_selftest_lane_shift": (i32, [vp, vp, C.c_float, vp]),
        "demod_kernel_ms": (C.c_double, [vp, C.POINTER(i32)]),
        "demod_selftest_q1_offsets": (i32, [vp, vp, C.c_uint64, vp]),
        "demod_last_call_launches": (i32, [vp]),
        "demod_enable_timing": (None, [vp, i32]),
        "demod_last_error_code": (i32, [vp]),
        "demod_set_head_mode": (i32, [vp, i32]),
        "demod_stream_depth": (i32, []),
        "demod_ring_slots_for": (u32, [u64]),
        "demod_ring_create": (i32, [vp, vp, u32, C.POINTER(vp)]),
        "demod_ring_destroy": (None, [vp]),
        "demod_ring_input": (vp, [vp]),
        "demod_ring_commit": (i32, [vp, u64, i32]),
        "demod_ring_collect": (i32, [vp, C.POINTER(vp), C.POINTER(u64)]),
        "demod_ring_release": (i32, [vp]),
        "demod_plan_shards": (i32, [u64, i32, u64, u64, u64, C.POINTER(Shard)]),
        "demod_plan_channels": (i32, [u64, u64, i32, u64, C.POINTER(Shard)]),
        "demod_multi_create": (i32, [C.POINTER(Params), C.POINTER(i32), i32, C.POINTER(vp)]),
        "demod_multi_destroy": (None, [vp]),
        "demod_multi_devices": (i32, [vp]),
        "demod_multi_context": (vp, [vp, i32]),
        "demod_multi_stream": (vp, [vp, i32]),
        "demod_multi_plan": (i32, [vp, u64, u64, C.POINTER(Shard)]),
        "demod_multi_process_device": (i32, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(Shard)]),
        "demod_multi_synchronize": (i32, [vp]),
        "demod_multi_halo_transport": (C.c_char_p, [vp]),
        "demod_multi_last_error": (C.c_char_p, [vp]),
        "demod_design_magnitude_bound": (C.c_double, [C.POINTER(Params)]),
        "demod_design_unpack_folded": (C.c_int, [C.POINTER(Params)]),
        "demod_set_unpack_fold": (C.c_int, [C.c_void_p, C.c_int]),
        "demod_set_stencil_form": (C.c_int, [C.c_void_p, C.c_int]),
        "demod_stencil_form": (C.c_int, [C.c_void_p]),
        "demod_design_stencil_form": (C.c_int, [C.POINTER(Params), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "demod_design_normalized_constants": (C.c_int, [C.POINTER(Params), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                                       C.c_void_p, C.c_int]),
    }
    experiment = bool(os.environ.get("DEMOD_HIP_LIBRARY"))
    for name in EXPORTS:
        try:
            fn = getattr(lib, name)  # AttributeError here = the library does not match the header
        except AttributeError:
            if experiment:           # an older build selected on purpose for an A/B run (tools/)
                continue
            raise
        fn.restype, fn.argtypes = sig[name]
    if lib.demod_abi_version() != ABI_VERSION and not experiment:
        raise DemodError("libdemod_hip.so ABI version mismatch")
    _lib = lib
    return lib
