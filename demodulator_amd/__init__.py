"""demodulator_amd -- MI355X-native FM-demodulation hot path.

Host-side mirror of the reference's interface for this path: the options are
the reference CLI's (-S -L -l -D -d -e -m -b -q, /root/reference/src/main.c:125-183),
the work is done by hand-written HIP kernels behind the C ABI of
include/demod_hip.h.  PyTorch is used only as plumbing (device memory, streams,
torch.distributed); no compute happens in Python.
"""
from .demodulator import Demodulator, MultiDemodulator, Options, Ring, design, plan_shards_c, DEFAULT_BUF_SIZE  # noqa: F401
from ._native import DemodError, LIB_PATH, HALO_BYTES, ROW_BYTES  # noqa: F401

__all__ = ["Demodulator", "MultiDemodulator", "Ring", "plan_shards_c", "Options", "design", "DemodError", "DEFAULT_BUF_SIZE", "LIB_PATH", "HALO_BYTES",
           "ROW_BYTES"]
