"""h264-lab on MI355X: the reference's H264E_* encoder API served by hand-written HIP kernels (gfx950).

Python is only a thin ctypes mirror of the C API in include/h264e_mi355x.h, for tests and bench.py; the
product is lib/libh264e_mi355x.so (+ lib/encode_app).  There is no CPU fallback: loading fails loudly when
the HIP library has not been built, and encoder creation fails when no HIP device is present.
"""
from .binding import (  # noqa: F401
    CreateParam, RunParam, IoYuv, Encoder, ClipEncoder, ClipParam, ClipStats, load, lib_path, build, H264EError,
)
from .shard import StreamShard, shard_ranges  # noqa: F401
