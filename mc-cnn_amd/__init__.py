"""mc-cnn_amd: MI355X-native (gfx950) stereo matching-cost pipeline.

Drop-in for the `adcensus.*` operator table and `stereo_predict` of
jzbontar/mc-cnn (main.lua:929-1082), backed by hand-written HIP kernels in
csrc/ behind the C ABI of include/mc_adcensus.h (libmcadcensus.so).

The host side here is Python only because the reference's host toolchain
(LuaJIT + Torch7) is absent from this image; lua/adcensus.lua is the FFI shim
a Torch7 host would load instead.
"""
from . import _lib  # noqa: F401
from . import adcensus  # noqa: F401
from . import batch  # noqa: F401
from .binio import read_bin, write_bin  # noqa: F401
from .params import NET_SHAPES, PRESETS, TABLES, make_params  # noqa: F401
from .predict import stereo_predict, stereo_predict_fused, workspace_bytes  # noqa: F401

__all__ = ["adcensus", "batch", "stereo_predict", "stereo_predict_fused", "workspace_bytes", "read_bin", "write_bin",
           "PRESETS", "make_params"]
