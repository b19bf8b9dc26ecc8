"""Hyper-parameter tables of the reference (main.lua:68-295) as mc_params dicts.

Only the stereo-method flags that reach stereo_predict are kept:
L1 tau1 cbca_i1 cbca_i2 pi1 pi2 sgm_i sgm_q1 sgm_q2 alpha1 tau_so blur_sigma blur_t.
`lr_check` is the `dataset == 'kitti' or dataset == 'kitti2015'` branch
(main.lua:1054); `border_n` = (get_window_size(net)-1)/2 (main.lua:382-391,923):
4 conv layers of 3x3 -> 4 (kitti), 5 layers -> 5 (mb).
"""
import ctypes as C

def _t(L1, tau1, cbca_i1, cbca_i2, pi1, pi2, sgm_i, sgm_q1, sgm_q2, alpha1, tau_so, blur_sigma, blur_t, lr_check, border_n):
    return dict(L1=L1, tau1=tau1, cbca_i1=cbca_i1, cbca_i2=cbca_i2, pi1=pi1, pi2=pi2, sgm_i=sgm_i, sgm_q1=sgm_q1,
                sgm_q2=sgm_q2, alpha1=alpha1, tau_so=tau_so, blur_sigma=blur_sigma, blur_t=blur_t, lr_check=lr_check,
                border_n=border_n, median_k=5, sm_terminate=0, sm_skip=0)


# Every (dataset, arch) default table of main.lua:68-295.  lr_check = 1 for kitti / kitti2015 (main.lua:1054).
# border_n = (window - 1) / 2 of the net: l1 conv layers of 3x3 -> l1 (main.lua:382-391, 923); ad / census use no net.
TABLES = {
    ("kitti", "slow"): _t(5, 0.13, 2, 0, 1.32, 24.25, 1, 3.0, 2.0, 2.0, 0.08, 5.99, 6.0, 1, 4),        # main.lua:86-99
    ("kitti2015", "slow"): _t(5, 0.03, 2, 4, 2.3, 24.25, 1, 3.0, 2.0, 1.75, 0.08, 5.99, 5.0, 1, 4),    # main.lua:101-113
    ("mb", "slow"): _t(14, 0.02, 2, 16, 1.3, 13.9, 1, 4.5, 2.0, 2.75, 0.13, 1.67, 2.0, 0, 5),          # main.lua:132-144
    ("kitti", "census"): _t(0, 0.01, 4, 8, 4.0, 128.0, 1, 3.0, 3.5, 1.25, 1.0, 7.74, 6.0, 1, 0),       # main.lua:148-160
    ("kitti2015", "census"): _t(0, 0.01, 4, 8, 4.0, 128.0, 1, 3.0, 3.5, 1.25, 1.0, 7.74, 6.0, 1, 0),
    ("mb", "census"): _t(5, 0.22, 8, 8, 4.0, 32.0, 1, 4.0, 3.0, 1.5, 1.0, 2.78, 3.0, 0, 0),            # main.lua:162-174
    ("kitti", "ad"): _t(3, 0.03, 0, 4, 0.76, 13.93, 1, 3.5, 2.0, 2.5, 0.01, 7.74, 6.0, 1, 0),          # main.lua:178-190
    ("kitti2015", "ad"): _t(3, 0.03, 0, 4, 0.76, 13.93, 1, 3.5, 2.0, 2.5, 0.01, 7.74, 6.0, 1, 0),
    ("mb", "ad"): _t(5, 0.36, 0, 4, 0.4, 8.0, 1, 3.0, 4.0, 2.5, 0.08, 7.74, 1.0, 0, 0),                # main.lua:192-204
    ("kitti", "fast"): _t(0, 0.0, 0, 0, 4.0, 55.72, 1, 3.0, 2.5, 1.5, 0.02, 7.74, 5.0, 1, 4),          # main.lua:222-234
    ("kitti2015", "fast"): _t(0, 0.0, 0, 0, 2.3, 18.38, 1, 3.0, 2.0, 1.25, 0.08, 4.64, 5.0, 1, 4),     # main.lua:250-262
    ("mb", "fast"): _t(0, 0.0, 0, 0, 2.3, 24.3, 1, 4.0, 2.0, 1.5, 0.08, 6.0, 2.0, 0, 5),               # main.lua:281-293
}
# feature-net shapes (l1 conv layers of fm maps, 3x3): main.lua:73-75, 120-122, 212-214, 240-242, 270-272
NET_SHAPES = {("kitti", "fast"): (4, 64), ("kitti2015", "fast"): (4, 64), ("mb", "fast"): (5, 64),
              ("kitti", "slow"): (4, 112), ("kitti2015", "slow"): (4, 112), ("mb", "slow"): (5, 112)}

PRESETS = {"%s_%s" % k: v for k, v in TABLES.items()}


# -sm_terminate / -sm_skip stage names of main.lua:25-26 -> MC_SM_* / MC_SKIP_* (include/mc_adcensus.h)
SM_TERMINATE = {"": 0, "cnn": 1, "cbca1": 2, "sgm": 3, "cbca2": 4, "occlusion": 5, "mismatch": 6,
                "subpixel_enchancement": 7, "median": 8, "bilateral": 9}
SM_SKIP = {"": 0, "cbca": 1, "sgm": 2, "occlusion": 3, "subpixel_enchancement": 4, "median": 5, "bilateral": 6}


class McParams(C.Structure):
    """ctypes mirror of `mc_params` (include/mc_adcensus.h)."""
    _fields_ = [
        ("L1", C.c_int), ("tau1", C.c_float),
        ("cbca_i1", C.c_int), ("cbca_i2", C.c_int),
        ("pi1", C.c_float), ("pi2", C.c_float),
        ("sgm_i", C.c_int),
        ("sgm_q1", C.c_float), ("sgm_q2", C.c_float), ("alpha1", C.c_float), ("tau_so", C.c_float),
        ("blur_sigma", C.c_double), ("blur_t", C.c_float),
        ("lr_check", C.c_int), ("border_n", C.c_int), ("median_k", C.c_int),
        ("sm_terminate", C.c_int), ("sm_skip", C.c_int), ("left_only", C.c_int),
    ]


def make_params(d):
    if isinstance(d, McParams):
        return d
    if isinstance(d, str):
        d = PRESETS[d]
    p = McParams()
    for k, _ in McParams._fields_:
        v = d.get(k, 0) if k in ("sm_terminate", "sm_skip", "left_only") else d[k]
        if k == "sm_terminate" and isinstance(v, str):
            v = SM_TERMINATE[v]
        if k == "sm_skip" and isinstance(v, str):
            v = SM_SKIP[v]
        setattr(p, k, v)
    return p
