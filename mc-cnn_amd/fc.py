"""Accurate-architecture matching cost (arch slow): the per-disparity FC stack of main.lua:958-983 on the matrix cores.

`fc_cost_volumes(feat, layers, disp_max)` -> (volL, volR), each (1,D,H,W): NaN-filled (main.lua:966), with
net_te2(concat(featL[:,y,x], featR[:,y,x-d])) at volL[d,y,x] and volR[d,y,x-d] -- the raw volumes `stereo_predict`
continues from (fix_border is applied by the caller, main.lua:979).  `layers` = [(weight (out,in), bias (out)), ...] as
CUDA float tensors: the nn.SpatialConvolution1_fw weights of net_te2 (main.lua:688-695)."""
import ctypes as C

import torch

from ._lib import check, lib


def fc_cost_volumes(feat, layers, disp_max, workspace=None):
    feat = feat.contiguous()
    _, Cn, H, W = feat.shape
    D = int(disp_max)
    n = len(layers)
    ws = [w.contiguous() for w, _ in layers]
    bs = [b.contiguous().reshape(-1) for _, b in layers]
    for t in ws + bs + [feat]:
        if not (t.is_cuda and t.dtype == torch.float32):
            raise TypeError("contiguous float32 CUDA tensors expected")
    if ws[0].shape[1] != 2 * Cn:
        raise ValueError("layer 0 expects %d inputs, features give %d" % (ws[0].shape[1], 2 * Cn))
    need = lib.mc_fc_stack_workspace_bytes(Cn, n, H, W)
    if workspace is None or workspace.numel() < need + 16:
        workspace = torch.empty(need + 16, dtype=torch.uint8, device=feat.device)
    wsp = workspace.data_ptr() + (-workspace.data_ptr()) % 16
    vl = torch.full((1, D, H, W), float("nan"), dtype=torch.float32, device=feat.device)
    vr = torch.full((1, D, H, W), float("nan"), dtype=torch.float32, device=feat.device)
    wp = (C.c_void_p * n)(*[w.data_ptr() for w in ws])
    bp = (C.c_void_p * n)(*[b.data_ptr() for b in bs])
    widths = (C.c_int * n)(*[w.shape[0] for w in ws])
    check(lib.mc_fc_stack(feat[0].data_ptr(), feat[1].data_ptr(), Cn, H, W, D, wp, bp, widths, n, vl.data_ptr(), vr.data_ptr(),
                          wsp, need, torch.cuda.current_stream().cuda_stream), "fc_stack")
    return vl, vr
