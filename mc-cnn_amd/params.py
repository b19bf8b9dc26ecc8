"""Hyper-parameter tables of the reference (main.lua:68-295) as mc_params dicts.

Only the stereo-method flags that reach stereo_predict are kept:
L1 tau1 cbca_i1 cbca_i2 pi1 pi2 sgm_i sgm_q1 sgm_q2 alpha1 tau_so blur_sigma blur_t.
`lr_check` is the `dataset == 'kitti' or dataset == 'kitti2015'` branch
(main.lua:1054); `border_n` = (get_window_size(net)-1)/2 (main.lua:382-391,923):
4 conv layers of 3x3 -> 4 (kitti), 5 layers -> 5 (mb).
"""
import ctypes as C

PRESETS = {
    # main.lua:207-234
    "kitti_fast": dict(L1=0, tau1=0.0, cbca_i1=0, cbca_i2=0, pi1=4.0, pi2=55.72, sgm_i=1, sgm_q1=3.0, sgm_q2=2.5,
                       alpha1=1.5, tau_so=0.02, blur_sigma=7.74, blur_t=5.0, lr_check=1, border_n=4, median_k=5),
    # main.lua:86-99
    "kitti_slow": dict(L1=5, tau1=0.13, cbca_i1=2, cbca_i2=0, pi1=1.32, pi2=24.25, sgm_i=1, sgm_q1=3.0, sgm_q2=2.0,
                       alpha1=2.0, tau_so=0.08, blur_sigma=5.99, blur_t=6.0, lr_check=1, border_n=4, median_k=5),
    # main.lua:132-144
    "mb_slow": dict(L1=14, tau1=0.02, cbca_i1=2, cbca_i2=16, pi1=1.3, pi2=13.9, sgm_i=1, sgm_q1=4.5, sgm_q2=2.0,
                    alpha1=2.75, tau_so=0.13, blur_sigma=1.67, blur_t=2.0, lr_check=0, border_n=5, median_k=5),
}


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
    ]


def make_params(d):
    if isinstance(d, McParams):
        return d
    if isinstance(d, str):
        d = PRESETS[d]
    p = McParams()
    for k, _ in McParams._fields_:
        setattr(p, k, d[k])
    return p
