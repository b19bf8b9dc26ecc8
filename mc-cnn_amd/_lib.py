"""ctypes loader for libmcadcensus.so.  There is NO fallback: if the HIP library
is missing or fails to load, importing the package raises."""
import ctypes as C
import os

from .params import McParams

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmcadcensus.so")

# every symbol include/mc_adcensus.h declares
SYMBOLS = [
    "mc_version", "mc_last_error", "mc_fill_nan", "mc_stereo_join", "mc_ad", "mc_census_scratch_bytes", "mc_census_ws", "mc_fc_stack_workspace_bytes", "mc_fc_stack", "mc_conv3x3_workspace_bytes", "mc_conv3x3",
    "mc_fix_border",
    "mc_cross", "mc_cbca", "mc_cbca_scratch_bytes", "mc_cbca_ws", "mc_sgm2_tmp_bytes", "mc_sgm2", "mc_dhw_to_hwd", "mc_hwd_to_dhw", "mc_scale",
    "mc_argmin", "mc_spatial_argmin", "mc_outlier_detection", "mc_interpolate_occlusion",
    "mc_interpolate_mismatch", "mc_subpixel_enchancement", "mc_median2d", "mc_mean2d", "mc_gaussian_host",
    "mc_normalize_forward", "mc_predict_workspace_bytes", "mc_predict", "mc_predict_timed",
    "mc_cbca_plan_bytes", "mc_cbca_ws_cfg", "mc_transpose_cfg",
    "mc_read_png16", "mc_write_png16", "mc_write_pfm", "mc_grey2jet", "mc_sgm2_contract_violations",
]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "mc-cnn_amd: %s not found. Build it with `make -C mc-cnn_amd/csrc` (hipcc, gfx950) or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for s in SYMBOLS:
        getattr(lib, s)  # AttributeError if the ABI is incomplete
    lib.mc_last_error.restype = C.c_char_p
    lib.mc_sgm2_tmp_bytes.restype = C.c_size_t
    lib.mc_cbca_scratch_bytes.restype = C.c_size_t
    lib.mc_cbca_plan_bytes.restype = C.c_size_t
    lib.mc_census_scratch_bytes.restype = C.c_size_t
    lib.mc_fc_stack_workspace_bytes.restype = C.c_size_t
    lib.mc_predict_workspace_bytes.restype = C.c_size_t
    lib.mc_conv3x3_workspace_bytes.restype = C.c_size_t
    lib.mc_predict_workspace_bytes.argtypes = [C.POINTER(McParams), C.c_int, C.c_int, C.c_int, C.c_int]
    vp, i, f, i64, sz = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t
    sig = {
        "mc_fill_nan": [vp, i64, vp],
        "mc_stereo_join": [vp, vp, vp, vp, i, i, i, i, vp],
        "mc_ad": [vp, vp, vp, i, i, i, i, vp],
        "mc_census_scratch_bytes": [i, i, i],
        "mc_census_ws": [vp, vp, vp, i, i, i, i, i, vp, sz, vp],
        "mc_fc_stack_workspace_bytes": [i, i, i, i],
        "mc_fc_stack": [vp, vp, i, i, i, i, C.POINTER(vp), C.POINTER(vp), C.POINTER(i), i, vp, vp, vp, sz, vp],
        "mc_conv3x3_workspace_bytes": [i, i],
        "mc_conv3x3": [vp, vp, vp, vp, i, i, i, i, i, i, vp, sz, vp],
        "mc_fix_border": [vp, i, i, i, i, i, vp],
        "mc_cross": [vp, vp, i, i, i, f, vp],
        "mc_cbca": [vp, vp, vp, vp, i, i, i, i, vp],
        "mc_cbca_scratch_bytes": [i, i],
        "mc_cbca_plan_bytes": [i, i, i],
        "mc_cbca_ws": [vp, vp, vp, vp, i, i, i, i, vp, sz, vp],
        "mc_sgm2_tmp_bytes": [i, i, i],
        "mc_sgm2": [vp, vp, vp, vp, vp, sz, i, i, i, f, f, f, f, f, f, i, vp],
        "mc_sgm2_contract_violations": [vp, i, i, i, vp, vp],
        "mc_dhw_to_hwd": [vp, vp, i, i, i, vp],
        "mc_hwd_to_dhw": [vp, vp, i, i, i, f, vp],
        "mc_scale": [vp, vp, i64, f, vp],
        "mc_argmin": [vp, vp, i, i, i, vp],
        "mc_spatial_argmin": [vp, vp, i, i, i, vp],
        "mc_outlier_detection": [vp, vp, vp, i, i, i, vp],
        "mc_interpolate_occlusion": [vp, vp, vp, i, i, vp],
        "mc_interpolate_mismatch": [vp, vp, vp, i, i, vp],
        "mc_subpixel_enchancement": [vp, vp, vp, i, i, i, vp],
        "mc_median2d": [vp, vp, i, i, i, vp],
        "mc_mean2d": [vp, vp, vp, i, i, i, f, vp],
        "mc_gaussian_host": [C.c_double, vp, i],
        "mc_normalize_forward": [vp, vp, vp, i, i, i, i, vp],
        "mc_predict": [C.POINTER(McParams), vp, vp, vp, vp, i, vp, vp, i, i, i, vp, sz, vp, vp, vp, vp, vp, vp],
        "mc_predict_timed": [C.POINTER(McParams), vp, vp, vp, vp, i, vp, vp, i, i, i, vp, sz, vp, vp,
                             C.POINTER(C.c_float)],
        "mc_cbca_ws_cfg": [vp, vp, vp, vp, i, i, i, i, vp, sz, i, i, i, i, i, vp],
        "mc_transpose_cfg": [vp, vp, i64, i64, i64, i64, f, i, vp],
        "mc_read_png16": [C.c_char_p, vp, i64, C.POINTER(i), C.POINTER(i)],
        "mc_write_png16": [vp, i, i, C.c_char_p],
        "mc_write_pfm": [vp, i, i, C.c_char_p],
        "mc_grey2jet": [vp, vp, i, i],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        if name not in ("mc_sgm2_tmp_bytes", "mc_cbca_scratch_bytes", "mc_cbca_plan_bytes", "mc_census_scratch_bytes",
                        "mc_fc_stack_workspace_bytes", "mc_conv3x3_workspace_bytes"):
            fn.restype = C.c_int
    if lib.mc_version() != 8:
        raise ImportError("mc-cnn_amd: ABI version mismatch")
    return lib


lib = _load()


class McError(RuntimeError):
    """Raised where the reference raises a Lua error (checkCudaError, adcensus.cu:31-36)."""


def check(rc, what=""):
    if rc != 0:
        msg = lib.mc_last_error()
        raise McError("%s failed (rc=%d): %s" % (what or "libmcadcensus call", rc, msg.decode("utf-8", "replace") if msg else ""))
