"""-m gpu: the HIP path (C ABI) against the committed golden vectors of the reference
(tests/golden/*.npz, made by tests/golden/make_golden.py).  Bit-exact."""
import os

import numpy as np
import pytest

from util import diff_report, same_bits

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def same(got, want, name):
    got = got.detach().cpu().numpy() if hasattr(got, "detach") else got
    assert same_bits(got, want), diff_report(got, want, name)


@pytest.mark.parametrize("fname", ["ops_16x40x12.npz", "ops_9x36x20.npz"])
def test_ops(mc, fname):
    g = dict(np.load(os.path.join(GOLD, fname)))
    H, W, D, C = [int(v) for v in g["dims"]]
    A = mc.adcensus
    f = dev(g["feat"])
    vl = A.fill_nan(torch.empty((1, D, H, W), device="cuda"))
    vr = A.fill_nan(torch.empty((1, D, H, W), device="cuda"))
    A.StereoJoin(f[0], f[1], vl, vr)
    same(vl, g["join_L"], "StereoJoin left")
    same(vr, g["join_R"], "StereoJoin right")
    c0, c1 = np.stack([g["x0"], g["b0"]]), np.stack([g["x1"], g["b1"]])
    o = torch.empty((1, D, H, W), device="cuda")
    for direction, tag in ((-1, "m"), (1, "p")):
        A.ad(dev(g["x0"]), dev(g["x1"]), o, direction)
        same(o, g["ad_" + tag], "ad")
        A.census(dev(c0)[None], dev(c1)[None], o, direction)
        same(o, g["census_" + tag], "census")
    imgs = dict(s=g["x0"], s1=g["x1"], b=g["b0"], b1=g["b1"], z=g["x0"])
    arms = {}
    for (name, img), (L1, tau1) in zip(imgs.items(), g["cross_params"]):
        arms[name] = torch.empty((1, 4, H, W), device="cuda")
        A.cross(dev(img), arms[name], int(L1), float(tau1))
        same(arms[name], g["cross_" + name], "cross " + name)
    for direction, tag, vol in ((-1, "m", g["rawL"]), (1, "p", g["rawR"])):
        A.cbca(arms["s"], arms["s1"], dev(vol), o, direction)
        same(o, g["cbca_s_" + tag], "cbca smooth")
        A.cbca(arms["b"], arms["b1"], dev(vol), o, direction)
        same(o, g["cbca_b_" + tag], "cbca blocky")
    for i, prm in enumerate(g["sgm_params"]):
        for direction, tag, vol in ((-1, "m", g["rawL"]), (1, "p", g["rawR"])):
            vh = A.dhw_to_hwd(dev(vol)[None])
            out = torch.zeros((1, H, W, D), device="cuda")
            A.sgm2(dev(g["x0"]), dev(g["x1"]), vh, out, None, *[float(v) for v in prm], direction)
            same(out, g["sgm2_%d_%s" % (i, tag)], "sgm2")
    am = torch.empty((1, 1, H, W), device="cuda")
    A.spatial_argmin(dev(g["rawL"])[None], am)
    same(am, g["spatial_argmin_L"], "spatial_argmin")
    outl = torch.zeros((1, 1, H, W), device="cuda")
    A.outlier_detection(dev(g["d0"])[None, None], dev(g["d1"])[None, None], outl, D)
    same(outl, g["outlier"], "outlier")
    occ = A.interpolate_occlusion(dev(g["d0"])[None, None], outl)
    same(occ, g["occlusion"], "occlusion")
    mis = A.interpolate_mismatch(occ, outl)
    same(mis, g["mismatch"], "mismatch")
    sub = A.subpixel_enchancement(mis, dev(g["rawL"])[None], D)
    same(sub, g["subpixel"], "subpixel")
    med = A.median2d(sub, 5)
    same(med, g["median5"], "median")
    sigma, t = g["mean2d_params"]
    same(A.mean2d(med, A.gaussian(float(sigma)).cuda(), float(t)), g["mean2d"], "mean2d")
    x = dev(g["normalize_in"])
    nrm = torch.empty((2, 1, H, W), device="cuda")
    out = torch.empty_like(x)
    A.Normalize_forward(x, nrm, out)
    same(out, g["normalize_out"], "Normalize_forward")


@pytest.mark.parametrize("fname", ["predict_kitti_fast.npz", "predict_kitti_slow.npz", "predict_mb_slow.npz"])
@pytest.mark.parametrize("driver", ["fused", "ops"])
def test_stereo_predict(mc, fname, driver):
    g = dict(np.load(os.path.join(GOLD, fname)))
    H, W, D, C = [int(v) for v in g["dims"]]
    prm = {str(k): float(v) for k, v in zip(g["param_names"], g["param_values"])}
    for k in ("L1", "cbca_i1", "cbca_i2", "sgm_i", "lr_check", "border_n", "median_k"):
        prm[k] = int(prm[k])
    xb = dev(np.stack([g["x0"], g["x1"]]))[:, None]
    kw = dict(feat=dev(g["feat"])) if C else dict(raw=(dev(g["rawL"]), dev(g["rawR"])))
    if driver == "fused":
        got = mc.stereo_predict_fused(xb, prm, D, want_volumes=True, want_disp0=True, **kw)
    else:
        got = mc.stereo_predict(xb, prm, D, return_all=True, **kw)
    torch.cuda.synchronize()
    for k in ("volL", "volR", "dispL0", "dispR0", "disp"):
        same(got[k], g["out_" + k], k)
