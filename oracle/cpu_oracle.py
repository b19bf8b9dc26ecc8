"""ctypes/numpy binding of oracle/libmc_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (mc-cnn_amd/) never does.

Every function takes/returns C-contiguous float32 numpy arrays with the
reference's shapes ((D,H,W) volumes, (H,W,D) for sgm2, (4,H,W) arms).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmc_oracle.so")
_lib = None

_f32p = C.POINTER(C.c_float)


class OracleParams(C.Structure):
    """Mirror of `oracle_params` in mc_oracle.c."""
    _fields_ = [
        ("L1", C.c_int), ("tau1", C.c_float),
        ("cbca_i1", C.c_int), ("cbca_i2", C.c_int),
        ("pi1", C.c_float), ("pi2", C.c_float),
        ("sgm_i", C.c_int),
        ("sgm_q1", C.c_float), ("sgm_q2", C.c_float), ("alpha1", C.c_float), ("tau_so", C.c_float),
        ("blur_sigma", C.c_double), ("blur_t", C.c_float),
        ("lr_check", C.c_int), ("border_n", C.c_int), ("median_k", C.c_int),
        ("sm_terminate", C.c_int), ("sm_skip", C.c_int),
    ]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "mc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libmc_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_sgm2.restype = C.c_int
        _lib.oracle_gaussian.restype = C.c_int
        _lib.oracle_stereo_predict.restype = C.c_int
    return _lib


def _a(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    return x


def _p(x):
    return None if x is None else x.ctypes.data_as(_f32p)


def fill_nan(shape):
    out = np.empty(shape, np.float32)
    lib().oracle_fill_nan(_p(out), C.c_int64(out.size))
    return out


def stereo_join(featL, featR, D):
    featL, featR = _a(featL), _a(featR)
    Cn, H, W = featL.shape
    volL = fill_nan((D, H, W))
    volR = fill_nan((D, H, W))
    lib().oracle_stereo_join(_p(featL), _p(featR), _p(volL), _p(volR), Cn, D, H, W)
    return volL, volR


def ad(x0, x1, D, direction):
    x0, x1 = _a(x0), _a(x1)
    H, W = x0.shape[-2:]
    out = np.empty((D, H, W), np.float32)
    lib().oracle_ad(_p(x0), _p(x1), _p(out), D, H, W, direction)
    return out


def census(x0, x1, D, direction):
    x0, x1 = _a(x0), _a(x1)
    x0 = x0.reshape((-1,) + x0.shape[-2:])
    x1 = x1.reshape(x0.shape)
    Cimg, H, W = x0.shape
    out = np.empty((D, H, W), np.float32)
    lib().oracle_census(_p(x0), _p(x1), _p(out), Cimg, D, H, W, direction)
    return out


def argmin(vol):
    vol = _a(vol)
    D, H, W = vol.shape
    out = np.empty((H, W), np.float32)
    lib().oracle_argmin(_p(vol), _p(out), D, H, W)
    return out


def cross(img, L1, tau1):
    img = _a(img)
    H, W = img.shape
    out = np.empty((4, H, W), np.float32)
    lib().oracle_cross(_p(img), _p(out), H, W, int(L1), C.c_float(tau1))
    return out


def cbca(x0c, x1c, vol, direction):
    x0c, x1c, vol = _a(x0c), _a(x1c), _a(vol)
    D, H, W = vol.shape
    out = np.empty_like(vol)
    lib().oracle_cbca(_p(x0c), _p(x1c), _p(vol), _p(out), D, H, W, direction)
    return out


def sgm2(x0, x1, vol_hwd, pi1, pi2, tau_so, alpha1, q1, q2, direction, out=None):
    """Accumulates into `out` (zeros if None), like adcensus.sgm2."""
    x0, x1, vol_hwd = _a(x0), _a(x1), _a(vol_hwd)
    H, W, D = vol_hwd.shape
    if out is None:
        out = np.zeros_like(vol_hwd)
    rc = lib().oracle_sgm2(_p(x0), _p(x1), _p(vol_hwd), _p(out), H, W, D, C.c_float(pi1), C.c_float(pi2),
                           C.c_float(tau_so), C.c_float(alpha1), C.c_float(q1), C.c_float(q2), direction)
    if rc:
        raise ValueError("oracle_sgm2 rc=%d (D must be in 1..512)" % rc)
    return out


def outlier_detection(d0, d1, disp_max):
    d0, d1 = _a(d0), _a(d1)
    H, W = d0.shape
    out = np.empty_like(d0)
    lib().oracle_outlier_detection(_p(d0), _p(d1), _p(out), H, W, disp_max)
    return out


def interpolate_occlusion(d0, outlier):
    d0, outlier = _a(d0), _a(outlier)
    H, W = d0.shape
    out = np.empty_like(d0)
    lib().oracle_interpolate_occlusion(_p(d0), _p(outlier), _p(out), H, W)
    return out


def interpolate_mismatch(d0, outlier):
    d0, outlier = _a(d0), _a(outlier)
    H, W = d0.shape
    out = np.empty_like(d0)
    lib().oracle_interpolate_mismatch(_p(d0), _p(outlier), _p(out), H, W)
    return out


def subpixel_enchancement(d0, vol):
    d0, vol = _a(d0), _a(vol)
    D, H, W = vol.shape
    out = np.empty_like(d0)
    lib().oracle_subpixel_enchancement(_p(d0), _p(vol), _p(out), D, H, W)
    return out


def median2d(img, k):
    img = _a(img)
    H, W = img.shape
    out = np.empty_like(img)
    lib().oracle_median2d(_p(img), _p(out), H, W, k)
    return out


def gaussian(sigma):
    ks = lib().oracle_gaussian(C.c_double(sigma), None, 0)
    k = np.empty((ks, ks), np.float32)
    lib().oracle_gaussian(C.c_double(sigma), _p(k), ks * ks)
    return k


def mean2d(img, kernel, alpha2):
    img, kernel = _a(img), _a(kernel)
    H, W = img.shape
    out = np.empty_like(img)
    lib().oracle_mean2d(_p(img), _p(kernel), _p(out), H, W, kernel.shape[0], C.c_float(alpha2))
    return out


def fix_border(vol, n, direction):
    vol = _a(vol).copy()
    D, H, W = vol.shape
    lib().oracle_fix_border(_p(vol), D, H, W, n, direction)
    return vol


def dhw_to_hwd(vol):
    vol = _a(vol)
    D, H, W = vol.shape
    out = np.empty((H, W, D), np.float32)
    lib().oracle_dhw_to_hwd(_p(vol), _p(out), D, H, W)
    return out


def hwd_to_dhw(vol):
    vol = _a(vol)
    H, W, D = vol.shape
    out = np.empty((D, H, W), np.float32)
    lib().oracle_hwd_to_dhw(_p(vol), _p(out), D, H, W)
    return out


def normalize_forward(x):
    x = _a(x)
    N, Cn, H, W = x.shape
    out = np.empty_like(x)
    lib().oracle_normalize_forward(_p(x), _p(out), N, Cn, H, W)
    return out


def fc_stack(featL, featR, D, layers):
    """layers: [(W (out,in), b (out))...]; returns NaN-filled (D,H,W) left / right volumes with the valid voxels set."""
    featL, featR = _a(featL), _a(featR)
    Cn, H, W = featL.shape
    ws = [_a(w) for w, _ in layers]
    bs = [_a(b) for _, b in layers]
    n = len(layers)
    wp = (_f32p * n)(*[_p(w) for w in ws])
    bp = (_f32p * n)(*[_p(b) for b in bs])
    widths = (C.c_int * n)(*[w.shape[0] for w in ws])
    vl = np.full((D, H, W), np.nan, np.float32)
    vr = np.full((D, H, W), np.nan, np.float32)
    fn = lib().oracle_fc_stack
    fn.restype = None
    fn.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_f32p), C.POINTER(_f32p), C.POINTER(C.c_int),
                   C.c_int, _f32p, _f32p]
    fn(_p(featL), _p(featR), Cn, H, W, D, wp, bp, widths, n, _p(vl), _p(vr))
    return vl, vr


def grey2jet(grey):
    """adcensus.grey2jet, /root/reference/adcensus.cu:2000-2053 (host code in the reference; the debug images of main.lua:503,1242,1260):
    (H,W) float64 -> (3,H,W) float64, the five linear pieces of the jet map over val = 4 * grey, evaluated in doubles in the reference's
    expressions; raises ValueError where the reference asserts (val outside [-0.1, 4.1], NaN)."""
    g = np.ascontiguousarray(grey, np.float64)
    val = g * 4
    r, gg, b = np.zeros_like(val), np.zeros_like(val), np.zeros_like(val)
    p1 = (-0.1 <= val) & (val < 0.5)
    p2 = (0.5 <= val) & (val < 1.5)
    p3 = (1.5 <= val) & (val < 2.5)
    p4 = (2.5 <= val) & (val < 3.5)
    p5 = (3.5 <= val) & (val <= 4.1)
    if not (p1 | p2 | p3 | p4 | p5).all():
        raise ValueError("grey2jet: val outside [-0.1, 4.1] (adcensus.cu:2046-2047 asserts)")
    b[p1] = 0.5 + val[p1]                                   # adcensus.cu:2022-2025
    gg[p2] = val[p2] - 0.5; b[p2] = 1                       # :2026-2029
    r[p3] = val[p3] - 1.5; gg[p3] = 1; b[p3] = 1 - (val[p3] - 1.5)   # :2030-2033
    r[p4] = 1; gg[p4] = 1 - (val[p4] - 2.5)                 # :2034-2037
    r[p5] = 1 - (val[p5] - 3.5)                             # :2038-2041
    return np.stack([r, gg, b])


def make_params(d):
    p = OracleParams()
    term = {"": 0, "cnn": 1, "cbca1": 2, "sgm": 3, "cbca2": 4, "occlusion": 5, "mismatch": 6,
            "subpixel_enchancement": 7, "median": 8, "bilateral": 9}
    skip = {"": 0, "cbca": 1, "sgm": 2, "occlusion": 3, "subpixel_enchancement": 4, "median": 5, "bilateral": 6}
    for k, _ in OracleParams._fields_:
        v = d.get(k, 0) if k in ("sm_terminate", "sm_skip") else d[k]
        if isinstance(v, str):
            v = (term if k == "sm_terminate" else skip)[v]
        setattr(p, k, v)
    return p


def stereo_predict(params, x0, x1, D, featL=None, featR=None, rawL=None, rawR=None):
    """main.lua:929-1082.  Returns dict(volL, volR, dispL0, dispR0, outlier, disp)."""
    x0, x1 = _a(x0), _a(x1)
    H, W = x0.shape
    Cn = 0
    if featL is not None:
        featL, featR = _a(featL), _a(featR)
        Cn = featL.shape[0]
    else:
        rawL, rawR = _a(rawL), _a(rawR)
    p = make_params(params) if isinstance(params, dict) else params
    o = dict(volL=np.empty((D, H, W), np.float32), volR=np.empty((D, H, W), np.float32),
             dispL0=np.empty((H, W), np.float32), dispR0=np.empty((H, W), np.float32),
             outlier=np.empty((H, W), np.float32), disp=np.empty((H, W), np.float32))
    rc = lib().oracle_stereo_predict(C.byref(p), _p(x0), _p(x1), _p(featL), _p(featR), Cn, _p(rawL), _p(rawR),
                                     D, H, W, _p(o["volL"]), _p(o["volR"]), _p(o["dispL0"]), _p(o["dispR0"]),
                                     _p(o["outlier"]), _p(o["disp"]))
    if rc:
        raise ValueError("oracle_stereo_predict rc=%d" % rc)
    return o
