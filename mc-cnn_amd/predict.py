"""stereo_predict (main.lua:929-1082) on MI355X.

Two drivers over the same kernels:

* `stereo_predict`        -- the reference's call sequence op by op through the
  `adcensus.*` mirror (what main.lua does once the LuaJIT shim replaces
  libadcensus.so): (D,H,W) volumes, explicit transposes around sgm2, ping-pong
  tmp volumes for cbca.
* `stereo_predict_fused`  -- one C-ABI call (`mc_predict`) that runs the whole
  post-CNN pipeline on one stream inside a caller-owned workspace: StereoJoin
  writes (H,W,D) directly with NaN fill and fix_border folded in, the SGM sweeps
  fold the zeroing, the /4 and the arg-min, and nothing is copied in between.

Both take the cost-volume stage as input: `feat` = (2,C,H,W) normalised features
(arch fast) or `raw` = the two raw (D,H,W) volumes of arch slow / ad / census.
"""
import ctypes as C

import torch

from . import adcensus
from ._lib import check, lib
from .params import make_params


STAGES = ("prep", "join", "cbca", "layout", "sgm", "argmin", "post", "_")  # MC_STAGE_* of mc_adcensus.h


def _img2(x_batch):
    H, W = x_batch.shape[-2:]
    xb = x_batch.reshape(2, H, W)
    return xb[0].contiguous(), xb[1].contiguous(), H, W


def stereo_predict(x_batch, params, disp_max, feat=None, raw=None, return_all=False):
    """Op-by-op mirror of stereo_predict(x_batch, id), main.lua:929-1082.

    x_batch: (2,1,H,W) normalised images.  Returns disp[2] (1,1,H,W); with
    return_all also the final volumes (what left.bin / right.bin hold) and the
    intermediate maps."""
    p = make_params(params)
    x0, x1, H, W = _img2(x_batch)
    D = int(disp_max)
    dev = x_batch.device

    vols = [None, None]  # [0] = left (direction -1), [1] = right (+1): vols[{{direction == -1 and 1 or 2}}]
    if feat is not None:  # arch == 'fast', main.lua:944-951
        feat = feat.contiguous()
        vl = adcensus.fill_nan(torch.empty((1, D, H, W), dtype=torch.float32, device=dev))
        vr = adcensus.fill_nan(torch.empty((1, D, H, W), dtype=torch.float32, device=dev))
        adcensus.StereoJoin(feat[0], feat[1], vl, vr)
        adcensus.fix_border(vl, p.border_n, -1)
        adcensus.fix_border(vr, p.border_n, 1)
        vols = [vl, vr]
    else:
        vols = [raw[0].reshape(1, D, H, W).clone(), raw[1].reshape(1, D, H, W).clone()]

    disp = {}
    out_vols = {}
    for direction in (1, -1):  # main.lua:954-955
        vol = vols[0 if direction == -1 else 1]
        # main.lua:992-1004: cross is computed whenever cbca is not skipped, even for 0 iterations
        x0c = torch.empty((1, 4, H, W), dtype=torch.float32, device=dev)
        x1c = torch.empty((1, 4, H, W), dtype=torch.float32, device=dev)
        adcensus.cross(x0, x0c, p.L1, p.tau1)
        adcensus.cross(x1, x1c, p.L1, p.tau1)
        tmp_cbca = torch.empty_like(vol)
        for _ in range(p.cbca_i1):
            adcensus.cbca(x0c, x1c, vol, tmp_cbca, direction)
            vol, tmp_cbca = tmp_cbca, vol  # vol:copy(tmp_cbca)
        if p.sgm_i > 0:  # main.lua:1007-1030
            volh = adcensus.dhw_to_hwd(vol)
            out = torch.empty_like(volh)
            tmp = torch.empty((W, D), dtype=torch.float32, device=dev)
            for _ in range(p.sgm_i):
                out.zero_()
                adcensus.sgm2(x0, x1, volh, out, tmp, p.pi1, p.pi2, p.tau_so, p.alpha1, p.sgm_q1, p.sgm_q2, direction)
                adcensus.scale(out, volh, 0.25)  # vol:copy(out):div(4)
            vol = adcensus.hwd_to_dhw(out, 0.25, out=vol.reshape(1, D, H, W))  # main.lua:1019-1020
        tmp_cbca = torch.empty_like(vol)
        for _ in range(p.cbca_i2):  # main.lua:1033-1039
            adcensus.cbca(x0c, x1c, vol, tmp_cbca, direction)
            vol, tmp_cbca = tmp_cbca, vol
        out_vols[direction] = vol
        disp[1 if direction == 1 else 2] = adcensus.argmin(vol)  # main.lua:1049-1050

    d = disp[2]
    outlier = torch.zeros_like(d)
    if p.lr_check:  # main.lua:1054-1066
        adcensus.outlier_detection(d, disp[1], outlier, D)
        d = adcensus.interpolate_occlusion(d, outlier)
        d = adcensus.interpolate_mismatch(d, outlier)
    d = adcensus.subpixel_enchancement(d, out_vols[-1], D)  # vol = LEFT volume, main.lua:1068
    d = adcensus.median2d(d, p.median_k)
    d = adcensus.mean2d(d, adcensus.gaussian(p.blur_sigma).to(dev), p.blur_t)
    if return_all:
        return dict(disp=d, volL=out_vols[-1], volR=out_vols[1], dispL0=disp[2], dispR0=disp[1], outlier=outlier)
    return d


def workspace_bytes(params, disp_max, H, W, C_feat=0):
    p = make_params(params)
    n = lib.mc_predict_workspace_bytes(C.byref(p), int(C_feat), int(disp_max), int(H), int(W))
    if n == 0:
        raise ValueError("mc_predict_workspace_bytes: bad arguments")
    return n


class Workspace:
    """Caller-owned device scratch for mc_predict (one per concurrent stream)."""

    def __init__(self, params, disp_max, H, W, device):
        self.nbytes = workspace_bytes(params, disp_max, H, W)
        self.buf = torch.empty(self.nbytes + 256, dtype=torch.uint8, device=device)
        off = (-self.buf.data_ptr()) % 256
        self.ptr = self.buf.data_ptr() + off


def stereo_predict_fused(x_batch, params, disp_max, feat=None, raw=None, workspace=None, want_volumes=False,
                         want_disp0=False, out=None, timed=False):
    """The whole of stereo_predict from the cost-volume stage on, in one C-ABI call."""
    p = make_params(params)
    x0, x1, H, W = _img2(x_batch)
    D = int(disp_max)
    dev = x_batch.device
    if workspace is None:
        workspace = Workspace(p, D, H, W, dev)
    if out is None:
        out = torch.empty((1, 1, H, W), dtype=torch.float32, device=dev)
    fl = fr = rl = rr = None
    Cn = 0
    if feat is not None:
        feat = feat.contiguous()
        Cn = feat.shape[-3]
        fl, fr = feat[0].data_ptr(), feat[1].data_ptr()
        keep = feat
    else:
        keep = (raw[0].contiguous(), raw[1].contiguous())
        rl, rr = keep[0].data_ptr(), keep[1].data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    res = dict(disp=out)
    if timed:
        ms = (C.c_float * 8)()
        check(lib.mc_predict_timed(C.byref(p), x0.data_ptr(), x1.data_ptr(), fl, fr, Cn, rl, rr, D, H, W, workspace.ptr,
                                   workspace.nbytes, out.data_ptr(), st, ms), "mc_predict_timed")
        res["stage_ms"] = dict(zip(STAGES, list(ms)))
        return res
    vl = vr = dl = dr = None
    if want_volumes:
        res["volL"] = torch.empty((1, D, H, W), dtype=torch.float32, device=dev)
        res["volR"] = torch.empty((1, D, H, W), dtype=torch.float32, device=dev)
        vl, vr = res["volL"].data_ptr(), res["volR"].data_ptr()
    if want_disp0:
        res["dispL0"] = torch.empty((1, 1, H, W), dtype=torch.float32, device=dev)
        res["dispR0"] = torch.empty((1, 1, H, W), dtype=torch.float32, device=dev)
        dl, dr = res["dispL0"].data_ptr(), res["dispR0"].data_ptr()
    check(lib.mc_predict(C.byref(p), x0.data_ptr(), x1.data_ptr(), fl, fr, Cn, rl, rr, D, H, W, workspace.ptr,
                         workspace.nbytes, vl, vr, dl, dr, out.data_ptr(), st), "mc_predict")
    del keep
    return res
