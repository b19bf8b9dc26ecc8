"""The reference's `adcensus.*` operator table (adcensus.cu funcs[], 2061-2096) for
the hot path, with the same names, argument order and in-place / allocate-and-return
behaviour, over torch CUDA tensors.

Each function forwards raw device pointers and the dims the reference reads from
the SAME tensors (e.g. cross: H,W from `out`, adcensus.cu:336-337) to the C ABI
(include/mc_adcensus.h) on torch's current stream.  Like the reference, tensors
must be contiguous fp32 on the GPU; unlike it, that is checked.  A non-zero return
code raises McError where the reference raises a Lua error.
"""
import os

import torch

from ._lib import check, lib

_scratch = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*tensors):
    for t in tensors:
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise TypeError("torch.CudaTensor expected")  # luaT_checkudata failure
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("contiguous float32 CUDA tensor expected")


def _p(t):
    return t.data_ptr()


def _scratch_for(device, need):
    # one area per (device, size, stream): calls on different streams may run at the same time
    key = (device.index, need, _stream())
    scratch = _scratch.get(key)
    if scratch is None:
        scratch = _scratch[key] = torch.empty(need, dtype=torch.uint8, device=device)
    return scratch


def ad(x0, x1, out, direction):
    """adcensus.ad(x0, x1, out, direction) -- adcensus.cu:95-114."""
    _chk(x0, x1, out)
    D, H, W = out.shape[-3:]
    check(lib.mc_ad(_p(x0), _p(x1), _p(out), D, H, W, int(direction), _stream()), "ad")


def census(x0, x1, out, direction):
    """adcensus.census(x0, x1, out, direction) -- adcensus.cu:155-175 (channels = x0:size(2)).  Runs the signature
    form (mc_census_ws) with a cached per-shape scratch."""
    _chk(x0, x1, out)
    D, H, W = out.shape[-3:]
    cimg = x0.shape[-3] if x0.dim() >= 3 else 1
    need = lib.mc_census_scratch_bytes(cimg, H, W)
    scratch = _scratch_for(out.device, need)
    check(lib.mc_census_ws(_p(x0), _p(x1), _p(out), cimg, D, H, W, int(direction), scratch.data_ptr(), need, _stream()),
          "census")


def conv3x3(x, weight, bias, relu, out=None):
    """cudnn.SpatialConvolution(Cin, Cout, 3, 3, 1, 1, 1, 1) [+ ReLU] of net_te (main.lua:681-686, 727-746) through
    mc_conv3x3 (fp32 MFMA implicit GEMM).  x (N,Cin,H,W), weight (Cout,Cin,3,3), bias (Cout)."""
    _chk(x, weight, bias)
    N, Cin, H, W = x.shape
    Cout = weight.shape[0]
    if out is None:
        out = torch.empty((N, Cout, H, W), dtype=torch.float32, device=x.device)
    need = lib.mc_conv3x3_workspace_bytes(Cin, Cout)
    ws = _scratch_for(x.device, need)
    check(lib.mc_conv3x3(_p(x), _p(weight), _p(bias), _p(out), N, Cin, Cout, H, W, 1 if relu else 0, ws.data_ptr(), need,
                         _stream()), "conv3x3")
    return out


def StereoJoin(input_L, input_R, output_L, output_R):
    """adcensus.StereoJoin(input_L, input_R, output_L, output_R) -- adcensus.cu:1479-1498."""
    _chk(input_L, input_R, output_L, output_R)
    C = input_L.shape[-3]
    D, H, W = output_L.shape[-3:]
    check(lib.mc_stereo_join(_p(input_L), _p(input_R), _p(output_L), _p(output_R), C, D, H, W, _stream()),
          "StereoJoin")


def cross(x0, out, L1, tau1):
    """adcensus.cross(x0, out, L1, tau1) -- adcensus.cu:324-341."""
    _chk(x0, out)
    H, W = out.shape[-2:]
    check(lib.mc_cross(_p(x0), _p(out), H, W, int(L1), float(tau1), _stream()), "cross")


def cbca(x0c, x1c, vol_in, vol_out, direction):
    """adcensus.cbca(x0c, x1c, vol_in, vol_out, direction) -- adcensus.cu:379-400.  Runs the LDS-tiled kernel
    (mc_cbca_ws) with a cached per-shape scratch for the packed arm lengths."""
    _chk(x0c, x1c, vol_in, vol_out)
    D, H, W = vol_out.shape[-3:]
    need = lib.mc_cbca_scratch_bytes(H, W)
    scratch = _scratch_for(vol_out.device, need)
    check(lib.mc_cbca_ws(_p(x0c), _p(x1c), _p(vol_in), _p(vol_out), D, H, W, int(direction), scratch.data_ptr(), need,
                         _stream()), "cbca")


def cbca_cfg(x0c, x1c, vol_in, vol_out, direction, rb=0, nt=-1, d0=0, nd=0, form=0):
    """Test / bench hook (mc_cbca_ws_cfg): adcensus.cbca with the launch configuration forced -- non-temporal
    instantiation, plane range, kernel form (0 what adcensus.cbca does, 1 strip kernel with rb rows per strip, 2 / 3 tile
    kernel short-arm / long-arm instance with rb = geometry variant; these write nothing if an arm exceeds 4 / 13;
    4 / 5 tile kernel that also writes the pair's item order behind the packed lengths, 6 / 7 tile kernel that reads it:
    a 6 / 7 call must follow a 4 / 5 call on the same arms, shape and direction; 8 / 9 the lean kernel of textured pairs:
    8 first lists the outputs whose support is not the minimal 3 x 3 behind the packed lengths, 9 reads that list -- the strip
    kernel takes over if the list is not this problem's or did not fit; rb = rows per wave (2 / 4 / 8), d0 = launch variant, nd = slots
    the list may hold; 10 / 11 TWO aggregation passes in one launch (vol_out = the volume after the second): 10 writes the list of that
    kernel's wave geometry first, 11 reads it; rb = rows per wave (4 / 8 / 12), d0 > 0 = the list's cost limit in values per voxel)
    -- instead of derived from the problem."""
    _chk(x0c, x1c, vol_in, vol_out)
    D, H, W = vol_out.shape[-3:]
    need = lib.mc_cbca_scratch_bytes(H, W) + (lib.mc_cbca_plan_bytes(D, H, W) if form >= 4 else 0)
    if form >= 10:
        need += 512 + 4 * D * H * W
    scratch = _scratch_for(vol_out.device, need)
    check(lib.mc_cbca_ws_cfg(_p(x0c), _p(x1c), _p(vol_in), _p(vol_out), D, H, W, int(direction), scratch.data_ptr(), need,
                             int(rb), int(nt), int(d0), int(nd), int(form), _stream()), "cbca_cfg")


def cbca_cfg_list_header(device, D, H, W, form):
    """Test helper: the first eight words of the list the last cbca_cfg(form = 8 .. 11) call of this shape left behind the packed lengths
    -- (count, overflow, D, H, W, direction + 1, magic, rows per wave [+ 0x100 two-pass geometry]); overflow != 0 = the list was declared
    unusable and the strip kernel took the passes."""
    cs = lib.mc_cbca_scratch_bytes(H, W)
    need = cs + lib.mc_cbca_plan_bytes(D, H, W) + ((512 + 4 * D * H * W) if form >= 10 else 0)
    off = (cs + 255) // 256 * 256
    return _scratch_for(device, need)[off:off + 32].view(torch.int32).cpu().tolist()


def transpose_cfg(inp, out, rows, cols, ldin, ldout, scale=1.0, nt=-1):
    """Test hook (mc_transpose_cfg): the layout kernel with its cache policy forced."""
    _chk(inp, out)
    check(lib.mc_transpose_cfg(_p(inp), _p(out), int(rows), int(cols), int(ldin), int(ldout), float(scale), int(nt),
                               _stream()), "transpose_cfg")


def cbca_reference_shaped(x0c, x1c, vol_in, vol_out, direction):
    """The same operator through mc_cbca (one thread per voxel, no scratch)."""
    _chk(x0c, x1c, vol_in, vol_out)
    D, H, W = vol_out.shape[-3:]
    check(lib.mc_cbca(_p(x0c), _p(x1c), _p(vol_in), _p(vol_out), D, H, W, int(direction), _stream()), "cbca")


def sgm2_contract_violations(input):
    """Debug aid (no reference counterpart): the number of pixels of an (1,H,W,D) volume that violate sgm2's contract --
    d = 0 not finite, or a value that is not NaN behind a NaN (the reference's `<`-tree minimum and this library's
    fminf-recurrence agree on everything else).  Synchronises."""
    _chk(input)
    H, W, D = input.shape[-3:]
    cnt = torch.zeros(1, dtype=torch.int32, device=input.device)
    check(lib.mc_sgm2_contract_violations(_p(input), H, W, D, cnt.data_ptr(), _stream()), "sgm2_contract_violations")
    return int(cnt.item())


def sgm2(x0, x1, input, output, tmp, pi1, pi2, tau_so, alpha1, sgm_q1, sgm_q2, direction):
    """adcensus.sgm2(...) -- adcensus.cu:620-697.  input/output are (1,H,W,D); the four
    directional costs are added to `output`.  `tmp` is the reference's (W,D) line-state
    tensor; this implementation keeps line state in registers and needs a differently
    sized scratch (edge-class maps), so `tmp` is used only if it is large enough,
    otherwise a cached per-shape scratch is used."""
    _chk(x0, x1, input, output)
    H, W, D = input.shape[-3:]
    if os.environ.get("MC_CHECK_CONTRACTS") == "1":   # host-side debug switch; the library itself reads no environment
        bad = sgm2_contract_violations(input)
        if bad:
            raise ValueError("sgm2: %d pixels of the input volume violate the contract (d = 0 finite, NaNs form a tail in d)" % bad)
    need = lib.mc_sgm2_tmp_bytes(H, W, D)
    if isinstance(tmp, torch.Tensor) and tmp.is_cuda and tmp.is_contiguous() and tmp.numel() * tmp.element_size() >= need:
        scratch = tmp
    else:
        key = (input.device.index, need)
        scratch = _scratch.get(key)
        if scratch is None:
            scratch = _scratch[key] = torch.empty(need, dtype=torch.uint8, device=input.device)
    check(lib.mc_sgm2(_p(x0), _p(x1), _p(input), _p(output), scratch.data_ptr(), scratch.numel() * scratch.element_size(),
                      H, W, D, float(pi1), float(pi2), float(tau_so), float(alpha1), float(sgm_q1), float(sgm_q2),
                      int(direction), _stream()), "sgm2")


def spatial_argmin(input, output):
    """adcensus.spatial_argmin(input, output) -- adcensus.cu:264-278 (1-based index)."""
    _chk(input, output)
    D, H, W = input.shape[-3:]
    check(lib.mc_spatial_argmin(_p(input), _p(output), D, H, W, _stream()), "spatial_argmin")


def outlier_detection(d0, d1, outlier, disp_max):
    """adcensus.outlier_detection(d0, d1, outlier, disp_max) -- adcensus.cu:901-918."""
    _chk(d0, d1, outlier)
    H, W = d0.shape[-2:]
    check(lib.mc_outlier_detection(_p(d0), _p(d1), _p(outlier), H, W, int(disp_max), _stream()), "outlier_detection")


def interpolate_occlusion(d0, outlier):
    """adcensus.interpolate_occlusion(d0, outlier) -> new tensor -- adcensus.cu:1107-1125."""
    _chk(d0, outlier)
    out = torch.empty_like(d0)
    H, W = d0.shape[-2:]
    check(lib.mc_interpolate_occlusion(_p(d0), _p(outlier), _p(out), H, W, _stream()), "interpolate_occlusion")
    return out


def interpolate_mismatch(d0, outlier):
    """adcensus.interpolate_mismatch(d0, outlier) -> new tensor -- adcensus.cu:1060-1077."""
    _chk(d0, outlier)
    out = torch.empty_like(d0)
    H, W = d0.shape[-2:]
    check(lib.mc_interpolate_mismatch(_p(d0), _p(outlier), _p(out), H, W, _stream()), "interpolate_mismatch")
    return out


def subpixel_enchancement(d0, c2, disp_max):
    """adcensus.subpixel_enchancement(d0, c2, disp_max) -> new tensor -- adcensus.cu:1222-1239."""
    _chk(d0, c2)
    out = torch.empty_like(d0)
    H, W = d0.shape[-2:]
    check(lib.mc_subpixel_enchancement(_p(d0), _p(c2), _p(out), int(disp_max), H, W, _stream()), "subpixel_enchancement")
    return out


def median2d(img, kernel_size):
    """adcensus.median2d(img, kernel_size) -> new tensor -- adcensus.cu:1596-1613."""
    _chk(img)
    out = torch.empty_like(img)
    H, W = img.shape[-2:]
    check(lib.mc_median2d(_p(img), _p(out), H, W, int(kernel_size), _stream()), "median2d")
    return out


def mean2d(img, kernel, alpha2):
    """adcensus.mean2d(img, kernel, alpha2) -> new tensor -- adcensus.cu:1263-1282."""
    _chk(img, kernel)
    out = torch.empty_like(img)
    H, W = img.shape[-2:]
    check(lib.mc_mean2d(_p(img), _p(kernel), _p(out), H, W, kernel.shape[0], float(alpha2), _stream()), "mean2d")
    return out


def Normalize_forward(input, norm, output):
    """adcensus.Normalize_forward(input, norm, output) -- adcensus.cu:1310-1333."""
    _chk(input, norm, output)
    N, C, H, W = input.shape
    check(lib.mc_normalize_forward(_p(input), _p(norm), _p(output), N, C, H, W, _stream()), "Normalize_forward")


# ---- cutorch glue used by stereo_predict (not part of adcensus.* in the reference) ----

def fill_nan(t):
    """tensor:fill(0/0) -- main.lua:946."""
    _chk(t)
    check(lib.mc_fill_nan(_p(t), t.numel(), _stream()), "fill_nan")
    return t


def fix_border(vol, n, direction):
    """fix_border(net, vol, direction) with n = (get_window_size(net)-1)/2 -- main.lua:922-927."""
    _chk(vol)
    D, H, W = vol.shape[-3:]
    check(lib.mc_fix_border(_p(vol), D, H, W, int(n), int(direction), _stream()), "fix_border")


def dhw_to_hwd(vol):
    """vol:transpose(2,3):transpose(3,4):clone() -- main.lua:1008."""
    _chk(vol)
    D, H, W = vol.shape[-3:]
    out = torch.empty((1, H, W, D), dtype=torch.float32, device=vol.device)
    check(lib.mc_dhw_to_hwd(_p(vol), _p(out), D, H, W, _stream()), "dhw_to_hwd")
    return out


def hwd_to_dhw(vol_hwd, scale=1.0, out=None):
    """vol:copy(out:transpose(3,4):transpose(2,3)):div(4) -- main.lua:1019-1020 (scale = 0.25)."""
    _chk(vol_hwd)
    H, W, D = vol_hwd.shape[-3:]
    if out is None:
        out = torch.empty((1, D, H, W), dtype=torch.float32, device=vol_hwd.device)
    check(lib.mc_hwd_to_dhw(_p(vol_hwd), _p(out), D, H, W, float(scale), _stream()), "hwd_to_dhw")
    return out


def scale(src, dst, s):
    """dst:copy(src):mul(s) -- main.lua:1017 with s = 1/4."""
    _chk(src, dst)
    check(lib.mc_scale(_p(src), _p(dst), src.numel(), float(s), _stream()), "scale")
    return dst


def argmin(vol):
    """_, d = torch.min(vol, 2); d:add(-1) -- main.lua:1049-1050.  Returns (1,1,H,W) float."""
    _chk(vol)
    D, H, W = vol.shape[-3:]
    out = torch.empty((1, 1, H, W), dtype=torch.float32, device=vol.device)
    check(lib.mc_argmin(_p(vol), _p(out), D, H, W, _stream()), "argmin")
    return out


def gaussian(sigma):
    """gaussian(sigma) -- main.lua:528-540: host doubles, returned as a float32 CPU tensor."""
    import ctypes as C
    ks = lib.mc_gaussian_host(float(sigma), None, 0)
    if ks <= 0:
        check(ks, "gaussian")
    k = torch.empty((ks, ks), dtype=torch.float32)
    rc = lib.mc_gaussian_host(float(sigma), C.c_void_p(k.data_ptr()), ks * ks)
    if rc != ks:
        check(rc, "gaussian")
    return k


# ---- host side of libadcensus: ground-truth / submission image formats (torch.FloatTensor in the reference) -------------------

def readPNG16(img, fname):
    """adcensus.readPNG16(img, fname) -- adcensus.cu:1670-1686: fills the (H,W) float32 CPU tensor `img` with val / 256 (0 stays 0)."""
    import ctypes as C
    assert img.dtype == torch.float32 and img.device.type == "cpu" and img.is_contiguous(), "readPNG16: contiguous float32 CPU tensor expected"
    h, w = C.c_int(), C.c_int()
    check(lib.mc_read_png16(str(fname).encode(), C.c_void_p(img.data_ptr()), img.numel(), C.byref(h), C.byref(w)), "readPNG16")
    return h.value, w.value


def png16_size(fname):
    """(height, width) of a PNG file (mc_read_png16's size query): what a caller sizes the tensor for readPNG16 with."""
    import ctypes as C
    h, w = C.c_int(), C.c_int()
    check(lib.mc_read_png16(str(fname).encode(), None, 0, C.byref(h), C.byref(w)), "readPNG16")
    return h.value, w.value


def writePNG16(img, height, width, fname):
    """adcensus.writePNG16(img, height, width, fname) -- adcensus.cu:1688-1704."""
    import ctypes as C
    assert img.dtype == torch.float32 and img.device.type == "cpu" and img.is_contiguous() and img.numel() >= height * width
    check(lib.mc_write_png16(C.c_void_p(img.data_ptr()), int(height), int(width), str(fname).encode()), "writePNG16")


def writePFM(img, fname):
    """adcensus.writePFM(img, fname) -- adcensus.cu:1706-1721."""
    import ctypes as C
    assert img.dtype == torch.float32 and img.device.type == "cpu" and img.is_contiguous() and img.dim() == 2
    check(lib.mc_write_pfm(C.c_void_p(img.data_ptr()), img.shape[0], img.shape[1], str(fname).encode()), "writePFM")


def grey2jet(grey_img, col_img):
    """adcensus.grey2jet(grey_img, col_img) -- adcensus.cu:2000-2053 (host code there too; the debug images of main.lua:503,1242,1260):
    (H,W) float64 CPU tensor with values in [-0.025, 1.025] -> (..., 3, H, W) float64 CPU tensor, the jet colour map."""
    import ctypes as C
    assert grey_img.dtype == torch.float64 and grey_img.device.type == "cpu" and grey_img.is_contiguous() and grey_img.dim() == 2, \
        "grey2jet: contiguous (H,W) float64 CPU tensor expected"
    assert col_img.dtype == torch.float64 and col_img.device.type == "cpu" and col_img.is_contiguous()
    if 3 * grey_img.numel() != col_img.numel():
        raise ValueError("Size mismatch")   # (the reference's luaL_error, adcensus.cu:2008)
    check(lib.mc_grey2jet(C.c_void_p(grey_img.data_ptr()), C.c_void_p(col_img.data_ptr()), grey_img.shape[0], grey_img.shape[1]), "grey2jet")
