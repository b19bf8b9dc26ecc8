"""-m gpu: the reference's one real input pair (samples/input/kittiL.png / kittiR.png, committed as
tests/golden/kitti_sample_pair.npz) through the product.

  * BASELINE.json configs[0] -- the sample pair, disp_max 70, fast architecture (main.lua:207-234): CPU oracle == fused
    mc_predict == main.lua's stereo_predict over the reference's own kernels (oracle/_ref), bit for bit.  Features are
    seeded (no trained net is reachable), everything from the features on is the reference's path.
  * the KITTI-shaped accurate configuration (kitti_slow, disp_max 228, raw volumes) on the real arms: the tile kernel's
    short-arm instance on exactly the supports real images produce.
  * the Middlebury parameter set (L1 = 14, tau1 = 0.02, 2 + 16 iterations) on the pair mirror-tiled to 1000 x 1500, at 64
    disparities: the tile kernel's long-arm instance on real flat regions (clipped highlights, equal grey levels)."""
import os
import sys

import numpy as np
import pytest

from util import diff_report, features, raw_volumes, same_bits, sample_pair

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from test_gpu_fullsize import assert_same_bits_dev  # noqa: E402  (the `ref` fixture lives in conftest.py)


def test_baseline_config0_sample_pair_d70_fast(mc, oracle, ref):
    H, W, D, C = 370, 1226, 70, 64
    prm = dict(mc.PRESETS["kitti_fast"])
    x0, x1 = sample_pair()
    f = features(C, H, W, seed=42)
    want = oracle.stereo_predict(prm, x0, x1, D, featL=f[0], featR=f[1])
    xb = torch.from_numpy(np.stack([x0, x1])[:, None]).cuda()
    feat = torch.from_numpy(f).cuda()
    got = mc.stereo_predict_fused(xb, prm, D, feat=feat, want_volumes=True)
    torch.cuda.synchronize()
    for k in ("volL", "volR", "disp"):
        g = got[k].cpu().numpy()
        assert same_bits(g, want[k]), diff_report(g, want[k], "configs[0] %s: hip vs oracle" % k)
    from ref_pipeline import ref_stereo_predict
    r = ref_stereo_predict(ref, prm, xb, D, feat=feat)
    torch.cuda.synchronize()
    for k in ("volL", "volR", "disp"):
        assert_same_bits_dev(got[k], r[k], "configs[0] %s: hip vs the reference's kernels" % k)
    # and the op-by-op route an unchanged main.lua takes
    d2 = mc.stereo_predict(xb, prm, D, feat=feat)
    assert_same_bits_dev(d2, r["disp"], "configs[0] disp.bin, op-by-op route")


@pytest.mark.parametrize("preset,H,W,D", [("kitti_slow", 370, 1226, 228), ("mb_slow", 1000, 1500, 64)])
def test_accurate_configurations_on_the_real_arms(mc, ref, preset, H, W, D):
    from ref_pipeline import ref_stereo_predict
    from mc_cnn_amd.predict import Workspace
    prm = dict(mc.PRESETS[preset])
    device = torch.device("cuda", 0)
    x0, x1 = sample_pair(H, W)
    xb = torch.from_numpy(np.stack([x0, x1])[:, None]).to(device)
    vl, vr = raw_volumes(D, H, W, seed=11)
    kw = dict(raw=(torch.from_numpy(vl).to(device), torch.from_numpy(vr).to(device)))
    ws = Workspace(prm, D, H, W, device)
    got = mc.stereo_predict_fused(xb, prm, D, workspace=ws, want_volumes=True, want_disp0=True, **kw)
    torch.cuda.synchronize()
    want = ref_stereo_predict(ref, prm, xb, D, **kw)
    torch.cuda.synchronize()
    for key, label in (("volL", "left.bin"), ("volR", "right.bin"), ("dispL0", "left argmin"), ("dispR0", "right argmin"), ("disp", "disp.bin")):
        assert_same_bits_dev(got[key], want[key], "%s %s on the reference's sample pair" % (preset, label))
    # the op-by-op route (adcensus.cbca picks its kernel from the arms on the device) stays bit-exact on the same inputs
    d2 = mc.stereo_predict(xb, prm, D, **kw)
    assert_same_bits_dev(d2, want["disp"], "%s disp.bin, op-by-op route" % preset)
