"""Test helper: stereo_predict (main.lua:929-1082) transliterated statement by statement, with every
`adcensus.*` call going to the REFERENCE's own binding (oracle/_ref via oracle.ref_lib.RefLib) and
the cutorch tensor glue (fill(0/0), copies, transposes, :div(4)) done by torch -- all exact data
movement or one IEEE divide.  The single third-party arithmetic call, torch.min(vol, 2)
(main.lua:1049, cutorch), is replaced by the reference's in-repo restatement adcensus.spatial_argmin
(adcensus.cu:244-278, 1-based) -- see "parity unpinned" in DESIGN.md.
"""
import math

import torch


def gaussian(sigma):
    """main.lua:528-540 (Lua doubles -> torch.Tensor (double) -> :cuda() float)."""
    kr = math.ceil(sigma * 3)
    ks = kr * 2 + 1
    k = torch.empty((ks, ks), dtype=torch.float64)
    for i in range(1, ks + 1):
        for j in range(1, ks + 1):
            y = (i - 1) - kr
            x = (j - 1) - kr
            k[i - 1, j - 1] = math.exp(-(x * x + y * y) / (2 * sigma * sigma))
    return k


def fix_border(vol, n, direction):
    """main.lua:922-927; Lua index direction*i with negative = from the right end."""
    W = vol.shape[3]

    def col(li):  # 1-based Lua index (negative counts from the end) -> 0-based
        return li - 1 if li > 0 else W + li
    for i in range(1, n + 1):
        vol[:, :, :, col(direction * i)] = vol[:, :, :, col(direction * (n + 1))].clone()


def ref_stereo_predict(ref, prm, x_batch, disp_max, feat=None, raw=None):
    """x_batch (2,1,H,W) cuda.  feat (2,C,H,W) -> arch fast; raw = (left, right) (1,D,H,W) volumes."""
    dev = x_batch.device
    H, W = x_batch.shape[2:]
    D = disp_max
    nan = float("nan")
    if feat is not None:  # main.lua:944-951
        vols = torch.full((2, D, H, W), nan, dtype=torch.float32, device=dev)
        ref.call("StereoJoin", feat[0:1].contiguous(), feat[1:2].contiguous(), vols[0:1], vols[1:2])
        fix_border(vols[0:1], prm["border_n"], -1)
        fix_border(vols[1:2], prm["border_n"], 1)
    else:
        vols = torch.cat([raw[0].reshape(1, D, H, W), raw[1].reshape(1, D, H, W)]).clone()
    disp = {}
    out_vol = {}
    x0 = x_batch[0:1].contiguous()   # x_batch[1] in Lua: (1,H,W)
    x1 = x_batch[1:2].contiguous()
    vol = None
    for direction in (1, -1):  # main.lua:954-955
        vol = vols[(0 if direction == -1 else 1):(1 if direction == -1 else 2)].clone()
        x0c = torch.empty((1, 4, H, W), dtype=torch.float32, device=dev)
        x1c = torch.empty((1, 4, H, W), dtype=torch.float32, device=dev)
        ref.call("cross", x0, x0c, prm["L1"], prm["tau1"])
        ref.call("cross", x1, x1c, prm["L1"], prm["tau1"])
        tmp_cbca = torch.empty((1, D, H, W), dtype=torch.float32, device=dev)
        for _ in range(prm["cbca_i1"]):
            ref.call("cbca", x0c, x1c, vol, tmp_cbca, direction)
            vol.copy_(tmp_cbca)
        if prm["sgm_i"] > 0:
            vol = vol.transpose(1, 2).transpose(2, 3).clone().contiguous()  # (1,H,W,D)
            out = torch.empty((1, H, W, D), dtype=torch.float32, device=dev)
            tmp = torch.empty((W, D), dtype=torch.float32, device=dev)
            for _ in range(prm["sgm_i"]):
                out.zero_()
                ref.call("sgm2", x0, x1, vol, out, tmp, prm["pi1"], prm["pi2"], prm["tau_so"], prm["alpha1"],
                         prm["sgm_q1"], prm["sgm_q2"], direction)
                vol.copy_(out).div_(4)
            vol = out.transpose(2, 3).transpose(1, 2).contiguous().div_(4)  # (1,D,H,W)
        tmp_cbca = torch.empty((1, D, H, W), dtype=torch.float32, device=dev)
        for _ in range(prm["cbca_i2"]):
            ref.call("cbca", x0c, x1c, vol, tmp_cbca, direction)
            vol.copy_(tmp_cbca)
        out_vol[direction] = vol
        d1 = torch.empty((1, 1, H, W), dtype=torch.float32, device=dev)
        ref.call("spatial_argmin", vol, d1)      # stands in for torch.min(vol, 2), main.lua:1049
        disp[1 if direction == 1 else 2] = d1 - 1  # :add(-1), main.lua:1050
    res = dict(volL=out_vol[-1], volR=out_vol[1], dispL0=disp[2].clone(), dispR0=disp[1].clone())
    d = disp[2]
    if prm["lr_check"]:  # main.lua:1054-1066
        outlier = torch.zeros_like(d)
        ref.call("outlier_detection", d, disp[1], outlier, D)
        d = ref.call("interpolate_occlusion", d, outlier)[0]
        d = ref.call("interpolate_mismatch", d, outlier)[0]
        res["outlier"] = outlier
    d = ref.call("subpixel_enchancement", d, vol, D)[0]   # vol = LEFT volume (last loop iteration)
    d = ref.call("median2d", d, 5)[0]
    d = ref.call("mean2d", d, gaussian(prm["blur_sigma"]).float().to(dev), prm["blur_t"])[0]
    res["disp"] = d
    return res
