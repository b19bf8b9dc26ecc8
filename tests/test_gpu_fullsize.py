"""-m gpu: parity AT THE BENCHMARKED SHAPES.  The fused mc_predict -- exactly the call bench.py times, on
exactly bench.py's inputs (bench.make_inputs) -- against main.lua's stereo_predict run over the
REFERENCE'S OWN kernels (oracle/_ref, tests/ref_pipeline.py) on the same GPU:

  kitti_fast  370x1226x228 from 64-channel features        (BASELINE.json configs[1])
  kitti_slow  370x1226x228 from raw volumes, CBCA 2+0       (configs[2])
  mb_slow     1000x1500x256 from raw volumes, CBCA 2+16     (configs[3], the north-star shape)

left.bin / right.bin volumes, both arg-min maps and disp.bin, bit for bit, NaN masks included.  The
comparison runs on the device (the volumes are 0.4 - 1.5 GB each).  These are the launches whose
geometry depends on the problem size (rows per CBCA strip, non-temporal instantiations, SGM line
counts), which no small-shape test reaches (main.lua:929-1082, 1144-1148)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def assert_same_bits_dev(got, want, name):
    """bit-exact equality on the device, NaN == NaN, NaN masks identical"""
    got = got.reshape(-1)
    want = want.reshape(-1)
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, tuple(got.shape), tuple(want.shape))
    ng, nw = torch.isnan(got), torch.isnan(want)
    bad = (ng != nw) | (~ng & (got.view(torch.int32) != want.view(torch.int32)))
    n = int(bad.sum().item())
    if n:
        idx = torch.nonzero(bad)[:5, 0]
        raise AssertionError("%s: %d of %d elements differ; first idx=%s got=%s want=%s" % (
            name, n, got.numel(), idx.tolist(), got[idx].tolist(), want[idx].tolist()))


@pytest.mark.parametrize("config", ["kitti_fast", "kitti_slow", "mb_slow"])
def test_fused_predict_at_benchmarked_shape(ref, mc, config):
    import bench
    from ref_pipeline import ref_stereo_predict
    from mc_cnn_amd.predict import Workspace
    cfg = bench.CONFIGS[config]
    preset, H, W, D, C, _ = cfg
    prm = dict(mc.PRESETS[preset])
    device = torch.device("cuda", 0)
    xb, kw, _ = bench.make_inputs(cfg, 0, device)
    ws = Workspace(prm, D, H, W, device)
    got = mc.stereo_predict_fused(xb, prm, D, workspace=ws, want_volumes=True, want_disp0=True, **kw)
    torch.cuda.synchronize()
    want = ref_stereo_predict(ref, prm, xb, D, **kw)
    torch.cuda.synchronize()
    for key, label in (("volL", "left.bin"), ("volR", "right.bin"), ("dispL0", "left argmin"),
                       ("dispR0", "right argmin"), ("disp", "disp.bin")):
        assert_same_bits_dev(got[key], want[key], "%s %s: hip fused vs reference kernels" % (config, label))
    # the timed call (no volume export) must produce the same disp.bin
    out = torch.empty((1, 1, H, W), dtype=torch.float32, device=device)
    mc.stereo_predict_fused(xb, prm, D, workspace=ws, out=out, **kw)
    assert_same_bits_dev(out, want["disp"], "%s disp.bin (timed call form)" % config)
    # sanity on the reference side: the arg-min maps are integers in [0, D)
    for k in ("dispL0", "dispR0"):
        d = want[k]
        assert float(d.min()) >= 0 and float(d.max()) <= D - 1 and bool((d == d.round()).all())


@pytest.mark.parametrize("preset,H,W,D", [("mb_slow", 260, 700, 48), ("kitti2015_slow", 200, 640, 64), ("mb_census", 130, 500, 40)])
def test_fused_predict_on_real_scene_arm_statistics(ref, mc, preset, H, W, D):
    """the regime real images are in (tests/util.natural_pair): cbca by the tile kernel (short-arm instance for L1 <= 5, long-arm
    instance for L1 <= 14, picked by the pair's route word), both aggregation blocks, through mc_predict, against the reference's kernels"""
    from ref_pipeline import ref_stereo_predict
    from mc_cnn_amd.predict import Workspace
    from util import natural_pair, raw_volumes
    prm = dict(mc.PRESETS[preset])
    device = torch.device("cuda", 0)
    x0, x1 = natural_pair(H, W, D, seed=7, sigma=25.0)
    xb = torch.from_numpy(np.stack([x0, x1])[:, None]).to(device)
    vl, vr = raw_volumes(D, H, W, seed=11)
    kw = dict(raw=(torch.from_numpy(vl).to(device), torch.from_numpy(vr).to(device)))
    ws = Workspace(prm, D, H, W, device)
    got = mc.stereo_predict_fused(xb, prm, D, workspace=ws, want_volumes=True, want_disp0=True, **kw)
    torch.cuda.synchronize()
    want = ref_stereo_predict(ref, prm, xb, D, **kw)
    torch.cuda.synchronize()
    for key, label in (("volL", "left.bin"), ("volR", "right.bin"), ("dispL0", "left argmin"), ("dispR0", "right argmin"), ("disp", "disp.bin")):
        assert_same_bits_dev(got[key], want[key], "%s %s on the realistic pair" % (preset, label))
