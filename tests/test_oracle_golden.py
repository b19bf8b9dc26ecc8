"""CPU (-m "not gpu"): the oracle (oracle/mc_oracle.c) against golden vectors produced by the
REFERENCE ITSELF -- /root/reference/adcensus.cu compiled for gfx950 and run on an MI355X by
tests/golden/make_golden.py (committed next to the vectors).  Bit-exact, NaN masks included.
This is what pins the oracle when neither a GPU nor /root/reference is around."""
import os

import numpy as np
import pytest

from util import diff_report, same_bits

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def same(got, want, name):
    assert same_bits(got, want), diff_report(got, want, name)


@pytest.mark.parametrize("fname", ["ops_16x40x12.npz", "ops_9x36x20.npz"])
def test_ops_match_reference(oracle, fname):
    g = load(fname)
    H, W, D, C = [int(v) for v in g["dims"]]
    jl, jr = oracle.stereo_join(g["feat"][0], g["feat"][1], D)
    same(jl, g["join_L"], "StereoJoin left")
    same(jr, g["join_R"], "StereoJoin right")
    c0, c1 = np.stack([g["x0"], g["b0"]]), np.stack([g["x1"], g["b1"]])
    for direction, tag in ((-1, "m"), (1, "p")):
        same(oracle.ad(g["x0"], g["x1"], D, direction), g["ad_" + tag], "ad")
        same(oracle.census(c0, c1, D, direction), g["census_" + tag], "census")
    imgs = dict(s=g["x0"], s1=g["x1"], b=g["b0"], b1=g["b1"], z=g["x0"])
    arms = {}
    for (name, img), (L1, tau1) in zip(imgs.items(), g["cross_params"]):
        arms[name] = oracle.cross(img, int(L1), float(tau1))
        same(arms[name], g["cross_" + name], "cross " + name)
    for direction, tag, vol in ((-1, "m", g["rawL"]), (1, "p", g["rawR"])):
        same(oracle.cbca(arms["s"], arms["s1"], vol, direction), g["cbca_s_" + tag], "cbca smooth")
        same(oracle.cbca(arms["b"], arms["b1"], vol, direction), g["cbca_b_" + tag], "cbca blocky")
    for i, prm in enumerate(g["sgm_params"]):
        for direction, tag, vol in ((-1, "m", g["rawL"]), (1, "p", g["rawR"])):
            got = oracle.sgm2(g["x0"], g["x1"], oracle.dhw_to_hwd(vol), *[float(v) for v in prm], direction)
            same(got, g["sgm2_%d_%s" % (i, tag)], "sgm2")
    same(oracle.argmin(g["rawL"]) + 1, g["spatial_argmin_L"], "spatial_argmin")
    outl = oracle.outlier_detection(g["d0"], g["d1"], D)
    same(outl, g["outlier"], "outlier_detection")
    occ = oracle.interpolate_occlusion(g["d0"], outl)
    same(occ, g["occlusion"], "interpolate_occlusion")
    mis = oracle.interpolate_mismatch(occ, outl)
    same(mis, g["mismatch"], "interpolate_mismatch")
    sub = oracle.subpixel_enchancement(mis, g["rawL"])
    same(sub, g["subpixel"], "subpixel_enchancement")
    med = oracle.median2d(sub, 5)
    same(med, g["median5"], "median2d")
    sigma, t = g["mean2d_params"]
    same(oracle.mean2d(med, oracle.gaussian(float(sigma)), float(t)), g["mean2d"], "mean2d")
    same(oracle.normalize_forward(g["normalize_in"]), g["normalize_out"], "Normalize_forward")


@pytest.mark.parametrize("fname", ["predict_kitti_fast.npz", "predict_kitti_slow.npz", "predict_mb_slow.npz"])
def test_stereo_predict_matches_reference(oracle, fname):
    """stereo_predict (main.lua:929-1082) over the reference's kernels: left.bin / right.bin contents,
    both arg-min maps, disp.bin."""
    g = load(fname)
    H, W, D, C = [int(v) for v in g["dims"]]
    prm = {str(k): float(v) for k, v in zip(g["param_names"], g["param_values"])}
    for k in ("L1", "cbca_i1", "cbca_i2", "sgm_i", "lr_check", "border_n", "median_k"):
        prm[k] = int(prm[k])
    if C:
        got = oracle.stereo_predict(prm, g["x0"], g["x1"], D, featL=g["feat"][0], featR=g["feat"][1])
    else:
        got = oracle.stereo_predict(prm, g["x0"], g["x1"], D, rawL=g["rawL"], rawR=g["rawR"])
    for k in ("volL", "volR", "dispL0", "dispR0", "disp"):
        same(got[k], g["out_" + k], k)
