"""-m gpu: every HIP kernel, called through the C ABI (ctypes mirror in mc-cnn_amd/adcensus.py),
against the CPU oracle on the same seeded inputs.  Bar: BIT-EXACT, NaN masks included
(the path is min/add/compare plus order-preserved fp32 sums), for volumes, disparity
indices and the final sub-pixel map alike."""
import numpy as np
import pytest

from util import (blocky_pair, diff_report, features, random_pair, raw_volumes, same_bits, smooth_pair)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def assert_same(got, want, name):
    assert same_bits(got, want), diff_report(got, want, name)


SHAPES = [  # (H, W, D) incl. ragged: D % 4 != 0, D > 256 (8 per lane), H > W, W < D
    (24, 40, 16), (17, 33, 7), (9, 70, 70), (40, 24, 12), (12, 50, 64), (6, 300, 260), (5, 20, 33),
]


@pytest.mark.parametrize("H,W,D", SHAPES)
@pytest.mark.parametrize("C", [1, 64])
def test_stereo_join(mc, oracle, H, W, D, C):
    f = features(C, H, W, seed=H * W + C)
    want_l, want_r = oracle.stereo_join(f[0], f[1], D)
    fd = dev(f)
    vl = mc.adcensus.fill_nan(torch.empty((1, D, H, W), device="cuda"))
    vr = mc.adcensus.fill_nan(torch.empty((1, D, H, W), device="cuda"))
    mc.adcensus.StereoJoin(fd[0], fd[1], vl, vr)
    assert_same(host(vl), want_l, "volL")
    assert_same(host(vr), want_r, "volR")


@pytest.mark.parametrize("H,W,D", SHAPES[:4])
@pytest.mark.parametrize("direction", [-1, 1])
def test_ad_census(mc, oracle, H, W, D, direction):
    x0, x1 = random_pair(H, W, seed=3)
    out = torch.empty((1, D, H, W), device="cuda")
    mc.adcensus.ad(dev(x0), dev(x1), out, direction)
    assert_same(host(out), oracle.ad(x0, x1, D, direction), "ad")
    c0 = np.stack([x0, x1 * 0.5])
    c1 = np.stack([x1, x0 * 2.0])
    mc.adcensus.census(dev(c0)[None], dev(c1)[None], out, direction)
    assert_same(host(out), oracle.census(c0, c1, D, direction), "census (signatures, mc_census_ws)")


@pytest.mark.parametrize("H,W,D", [(37, 150, 40), (5, 9, 12), (20, 70, 80)])
@pytest.mark.parametrize("direction", [-1, 1])
def test_ad_census_ties_and_borders(mc, oracle, H, W, D, direction):
    """Quantised images (many equal intensities: `<` ties), windows larger than the image, W < D."""
    rng = np.random.default_rng(H * W)
    x0 = rng.integers(0, 4, (H, W)).astype(np.float32)
    x1 = rng.integers(0, 4, (H, W)).astype(np.float32)
    out = torch.empty((1, D, H, W), device="cuda")
    mc.adcensus.ad(dev(x0), dev(x1), out, direction)
    assert_same(host(out), oracle.ad(x0, x1, D, direction), "ad")
    mc.adcensus.census(dev(x0)[None, None], dev(x1)[None, None], out, direction)
    assert_same(host(out), oracle.census(x0[None], x1[None], D, direction), "census")


@pytest.mark.parametrize("H,W", [(24, 40), (17, 33), (40, 9), (1, 50), (30, 1)])
@pytest.mark.parametrize("L1,tau1", [(0, 0.0), (5, 0.13), (14, 0.02), (14, 1e9), (3, 0.5)])
def test_cross(mc, oracle, H, W, L1, tau1):
    for mk in (random_pair, blocky_pair):
        img, _ = mk(H, W, seed=5)
        out = torch.empty((1, 4, H, W), device="cuda")
        mc.adcensus.cross(dev(img), out, L1, tau1)
        assert_same(host(out), oracle.cross(img, L1, tau1), "cross")


@pytest.mark.parametrize("H,W,D", SHAPES[:5])
@pytest.mark.parametrize("L1,tau1", [(0, 0.0), (5, 0.4), (14, 1e9)])
@pytest.mark.parametrize("direction", [-1, 1])
def test_cbca(mc, oracle, H, W, D, L1, tau1, direction):
    x0, x1 = blocky_pair(H, W, seed=11)
    x0c, x1c = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=13)
    vol = vl if direction == -1 else vr
    want = oracle.cbca(x0c, x1c, vol, direction)
    out = torch.empty((1, D, H, W), device="cuda")
    mc.adcensus.cbca(dev(x0c), dev(x1c), dev(vol), out, direction)
    assert_same(host(out), want, "cbca (packed-arm strips, mc_cbca_ws)")
    out2 = torch.empty((1, D, H, W), device="cuda")
    mc.adcensus.cbca_reference_shaped(dev(x0c), dev(x1c), dev(vol), out2, direction)
    assert_same(host(out2), want, "cbca (mc_cbca)")


@pytest.mark.parametrize("H,W,D", [(50, 200, 40), (33, 131, 17), (16, 64, 8), (17, 65, 9), (70, 90, 100),
                                   (20, 253, 6), (45, 519, 5), (9, 1010, 3), (83, 254, 3), (3, 5, 2)])
@pytest.mark.parametrize("mk,L1,tau1", [("smooth", 14, 0.02), ("smooth", 14, 0.3), ("blocky", 14, 0.2), ("flat", 13, 1.0),
                                        ("flat", 30, 1.0), ("blocky", 40, 0.3), ("random", 5, 0.13)])
@pytest.mark.parametrize("direction", [-1, 1])
def test_cbca_strip_shapes(mc, oracle, H, W, D, mk, L1, tau1, direction):
    """Strip edges (W around multiples of 252, W % 4 != 0), row chunks (H % 40), D % 4 != 0, supports from the minimal 3x3
    through the window form to arms far beyond the LDS ring (in-launch global fallback)."""
    if mk == "smooth":
        x0, x1 = smooth_pair(H, W, min(D, 8), seed=H)
    elif mk == "blocky":
        x0, x1 = blocky_pair(H, W, seed=W)
    elif mk == "random":
        x0, x1 = random_pair(H, W, seed=3)
    else:
        x0 = np.zeros((H, W), np.float32)
        x1 = np.zeros((H, W), np.float32)
        x1[H // 2:, W // 3:] = 2.0  # one edge so that left and right arms differ
    x0c, x1c = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=13)
    vol = vl if direction == -1 else vr
    want = oracle.cbca(x0c, x1c, vol, direction)
    out = torch.empty((1, D, H, W), device="cuda")
    mc.adcensus.cbca(dev(x0c), dev(x1c), dev(vol), out, direction)
    assert_same(host(out), want, "cbca strips")


SGM_PARAMS = [  # pi1, pi2, tau_so, alpha1, q1, q2
    (4.0, 55.72, 0.02, 1.5, 3.0, 2.5),
    (1.32, 24.25, 0.08, 2.0, 3.0, 2.0),
    (1.3, 13.9, 0.13, 2.75, 4.5, 2.0),
]


@pytest.mark.parametrize("H,W,D", SHAPES)
@pytest.mark.parametrize("prm", SGM_PARAMS)
@pytest.mark.parametrize("direction", [-1, 1])
def test_sgm2(mc, oracle, H, W, D, prm, direction):
    if H > W:
        pytest.skip("reference tmp indexing aliases for H > W (SURVEY 5); covered by test_sgm2_portrait")
    x0, x1 = smooth_pair(H, W, min(D, 8), seed=21)
    vl, vr = raw_volumes(D, H, W, seed=23)
    vol = oracle.dhw_to_hwd(vl if direction == -1 else vr)
    want = oracle.sgm2(x0, x1, vol, *prm, direction)
    out = torch.zeros((1, H, W, D), device="cuda")
    tmp = torch.empty((W, D), device="cuda")
    mc.adcensus.sgm2(dev(x0), dev(x1), dev(vol)[None], out, tmp, *prm, direction)
    assert_same(host(out), want, "sgm2")


def test_sgm2_portrait(mc, oracle):
    """H > W: the oracle strides its line state by max(H,W) (the reference would alias)."""
    H, W, D = 40, 24, 12
    x0, x1 = smooth_pair(H, W, 6, seed=2)
    vl, _ = raw_volumes(D, H, W, seed=3)
    vol = oracle.dhw_to_hwd(vl)
    want = oracle.sgm2(x0, x1, vol, *SGM_PARAMS[0], -1)
    out = torch.zeros((1, H, W, D), device="cuda")
    mc.adcensus.sgm2(dev(x0), dev(x1), dev(vol)[None], out, None, *SGM_PARAMS[0], -1)
    assert_same(host(out), want, "sgm2 portrait")


def test_sgm2_accumulates(mc, oracle):
    """adcensus.sgm2 ADDS into `output` (adcensus.cu:569,616)."""
    H, W, D = 10, 30, 8
    x0, x1 = smooth_pair(H, W, 4, seed=4)
    vl, _ = raw_volumes(D, H, W, seed=5)
    vol = oracle.dhw_to_hwd(vl)
    base = np.random.default_rng(0).random((H, W, D), dtype=np.float32)
    want = oracle.sgm2(x0, x1, vol, *SGM_PARAMS[1], -1, out=base.copy())
    out = dev(base)[None].clone()
    mc.adcensus.sgm2(dev(x0), dev(x1), dev(vol)[None], out, None, *SGM_PARAMS[1], -1)
    assert_same(host(out), want, "sgm2 accumulate")


@pytest.mark.parametrize("H,W,D", SHAPES[:5] + [(12, 64, 8), (30, 130, 20), (16, 100, 68)])   # (the last three: H*W and D multiples of 4, the 16-byte kernels)
def test_transposes_argmin(mc, oracle, H, W, D):
    vl, _ = raw_volumes(D, H, W, seed=31)
    vl[:, 0, 0] = np.nan  # all-NaN pixel -> index 0
    vl[3 % D, 1, 1] = vl[:, 1, 1][~np.isnan(vl[:, 1, 1])].min()  # tie -> first index
    d = dev(vl)[None]
    hwd = mc.adcensus.dhw_to_hwd(d)
    assert_same(host(hwd), oracle.dhw_to_hwd(vl), "dhw_to_hwd")
    back = mc.adcensus.hwd_to_dhw(hwd, 0.25)
    assert_same(host(back), vl / 4, "hwd_to_dhw/4")
    assert_same(host(mc.adcensus.argmin(d)), oracle.argmin(vl), "argmin")
    out = torch.empty((1, 1, H, W), device="cuda")
    mc.adcensus.spatial_argmin(d, out)
    assert_same(host(out), oracle.argmin(vl) + 1, "spatial_argmin")


def _disp_maps(H, W, D, seed):
    rng = np.random.default_rng(seed)
    from scipy.ndimage import gaussian_filter
    base = gaussian_filter(rng.random((H, W)), 4.0)
    base = (base - base.min()) / (np.ptp(base) + 1e-9) * (D - 1)
    d0 = np.floor(base).astype(np.float32)
    d1 = np.floor(np.roll(base, -3, axis=1)).astype(np.float32)
    noise = rng.random((H, W)) < 0.15
    d0[noise] = rng.integers(0, D, size=int(noise.sum())).astype(np.float32)
    return d0, d1


@pytest.mark.parametrize("H,W,D", [(24, 40, 16), (17, 33, 7), (40, 24, 12), (31, 130, 70)])
def test_post_chain(mc, oracle, H, W, D):
    d0, d1 = _disp_maps(H, W, D, seed=H + W)
    outl = torch.empty((1, 1, H, W), device="cuda")
    mc.adcensus.outlier_detection(dev(d0)[None, None], dev(d1)[None, None], outl, D)
    want_o = oracle.outlier_detection(d0, d1, D)
    assert_same(host(outl), want_o, "outlier")
    occ = mc.adcensus.interpolate_occlusion(dev(d0)[None, None], outl)
    want_occ = oracle.interpolate_occlusion(d0, want_o)
    assert_same(host(occ), want_occ, "occlusion")
    mis = mc.adcensus.interpolate_mismatch(occ, outl)
    want_mis = oracle.interpolate_mismatch(want_occ, want_o)
    assert_same(host(mis), want_mis, "mismatch")
    vl, _ = raw_volumes(D, H, W, seed=41)
    sub = mc.adcensus.subpixel_enchancement(mis, dev(vl)[None], D)
    want_sub = oracle.subpixel_enchancement(want_mis, vl)
    assert_same(host(sub), want_sub, "subpixel")
    for k in (1, 3, 5, 11):
        med = mc.adcensus.median2d(sub, k)
        assert_same(host(med), oracle.median2d(want_sub, k), "median%d" % k)
    med = mc.adcensus.median2d(sub, 5)
    want_med = oracle.median2d(want_sub, 5)
    for sigma, t in ((1.0, 2.0), (1.67, 2.0), (2.78, 3.0), (4.64, 5.0), (5.99, 6.0), (7.74, 5.0)):   # 1.0: the run-time-size kernel
        k = mc.adcensus.gaussian(sigma)
        assert_same(k.numpy(), oracle.gaussian(sigma), "gaussian")
        got = mc.adcensus.mean2d(med, k.cuda(), t)
        assert_same(host(got), oracle.mean2d(want_med, oracle.gaussian(sigma), t), "mean2d sigma=%g" % sigma)


def test_mismatch_all_outliers(mc, oracle):
    """every ray leaves the image: defined as 'keep d0' on both sides (reference: uninitialised read)."""
    H, W = 6, 9
    d0 = np.arange(H * W, dtype=np.float32).reshape(H, W)
    outl = np.full((H, W), 2, np.float32)
    got = mc.adcensus.interpolate_mismatch(dev(d0)[None, None], dev(outl)[None, None])
    assert_same(host(got), oracle.interpolate_mismatch(d0, outl), "mismatch all-2")


@pytest.mark.parametrize("H,W,p_mis", [(37, 53, 0.9), (64, 130, 0.97), (5, 300, 0.8), (201, 7, 0.95), (90, 121, 0.5)])
def test_mismatch_long_walks(mc, oracle, H, W, p_mis):
    """Mostly-mismatch marks: the 16 rays walk far, through the half-pixel rounding on both sides of zero and out of every
    image edge (the kernel walks in integer half pixels; the oracle accumulates floats and rounds like the reference)."""
    rng = np.random.default_rng(H * 1000 + W)
    d0 = rng.integers(0, 60, (H, W)).astype(np.float32) + rng.integers(0, 4, (H, W)).astype(np.float32) * 0.25
    outl = np.where(rng.random((H, W)) < p_mis, 2, rng.integers(0, 2, (H, W))).astype(np.float32)
    got = mc.adcensus.interpolate_mismatch(dev(d0)[None, None], dev(outl)[None, None])
    assert_same(host(got), oracle.interpolate_mismatch(d0, outl), "mismatch long walks")


def test_sgm2_contract_check(mc, monkeypatch):
    """mc_sgm2's documented input contract, checkable: volumes the pipeline produces pass; a NaN before a number, or a
    non-finite d = 0, is counted -- and the Python mirror refuses such a volume under MC_CHECK_CONTRACTS=1."""
    H, W, D = 9, 21, 12
    vl, _ = raw_volumes(D, H, W, seed=3)                       # (D,H,W) with the NaN triangle of a left volume
    hwd = np.ascontiguousarray(vl.transpose(1, 2, 0))[None]
    assert mc.adcensus.sgm2_contract_violations(dev(hwd)) == 0
    bad = hwd.copy()
    bad[0, 2, 5, 3] = np.nan                                   # a hole: finite values behind it
    bad[0, 4, 7, 0] = np.inf                                   # d = 0 not finite
    bad[0, 4, 8, 0] = np.nan
    assert mc.adcensus.sgm2_contract_violations(dev(bad)) == 3
    x = torch.zeros((1, 1, H, W), device="cuda")
    out = torch.zeros_like(dev(bad))
    monkeypatch.setenv("MC_CHECK_CONTRACTS", "1")
    with pytest.raises(ValueError, match="violate the contract"):
        mc.adcensus.sgm2(x, x, dev(bad), out, torch.empty(1, device="cuda"), 1.0, 8.0, 0.1, 2.0, 3.0, 2.0, -1)
    mc.adcensus.sgm2(x, x, dev(hwd), out, torch.empty(1, device="cuda"), 1.0, 8.0, 0.1, 2.0, 3.0, 2.0, -1)


def test_normalize_fix_border(mc, oracle):
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 16, 11, 23)).astype(np.float32)
    out = torch.empty_like(dev(x))
    norm = torch.empty((2, 1, 11, 23), device="cuda")
    mc.adcensus.Normalize_forward(dev(x), norm, out)
    assert_same(host(out), oracle.normalize_forward(x), "normalize")
    vl, vr = raw_volumes(12, 11, 23, seed=1)
    for vol, direction in ((vl, -1), (vr, 1)):
        for n in (0, 1, 4, 5):
            t = dev(vol)[None].clone()
            mc.adcensus.fix_border(t, n, direction)
            assert_same(host(t), oracle.fix_border(vol, n, direction), "fix_border")


def test_cbca_arms_beyond_packed_range(mc, oracle):
    """Arms longer than 254 pixels saturate the packed byte lengths: the standalone operator must notice and take
    the float-arm kernel (mc_cbca_ws's overflow flag), still bit-exact."""
    H, W, D = 3, 700, 3
    x0 = np.zeros((H, W), np.float32)
    x1 = np.zeros((H, W), np.float32)
    x0c, x1c = oracle.cross(x0, 400, 1.0), oracle.cross(x1, 400, 1.0)
    assert (np.arange(W)[None, :] - x0c[0] - 1).max() > 254
    vl, _ = raw_volumes(D, H, W, seed=3)
    want = oracle.cbca(x0c, x1c, vl, -1)
    out = torch.empty((1, D, H, W), device="cuda")
    mc.adcensus.cbca(dev(x0c), dev(x1c), dev(vl), out, -1)
    assert_same(host(out), want, "cbca with arms > 254")


PRED_CASES = [
    # name, preset overrides, H, W, D, C (0 = from raw volumes)
    ("kitti_fast", {}, 32, 96, 24, 64),
    ("kitti_fast", {}, 21, 70, 30, 16),          # D % 4 != 0
    ("kitti_fast", {"sgm_i": 2}, 16, 64, 16, 8),
    ("kitti_slow", {}, 32, 96, 24, 0),
    ("kitti_slow", {"cbca_i2": 1}, 20, 60, 18, 0),
    ("mb_slow", {"cbca_i2": 3}, 28, 80, 20, 0),
    ("mb_slow", {"cbca_i2": 2}, 28, 80, 20, 32),  # features through the (D,H,W) + cbca route
    ("kitti_fast", {}, 8, 300, 260, 4),           # D > 256: 8 disparities per lane
]


@pytest.mark.parametrize("name,over,H,W,D,C", PRED_CASES)
@pytest.mark.parametrize("driver", ["fused", "ops"])
def test_stereo_predict(mc, oracle, name, over, H, W, D, C, driver):
    prm = dict(mc.PRESETS[name])
    prm.update(over)
    x0, x1 = smooth_pair(H, W, min(D, 12), seed=77)
    xb = dev(np.stack([x0, x1]))[:, None]
    if C:
        f = features(C, H, W, seed=5)
        want = oracle.stereo_predict(prm, x0, x1, D, featL=f[0], featR=f[1])
        kw = dict(feat=dev(f))
    else:
        vl, vr = raw_volumes(D, H, W, seed=7)
        want = oracle.stereo_predict(prm, x0, x1, D, rawL=vl, rawR=vr)
        kw = dict(raw=(dev(vl), dev(vr)))
    if driver == "fused":
        got = mc.stereo_predict_fused(xb, prm, D, want_volumes=True, want_disp0=True, **kw)
    else:
        got = mc.stereo_predict(xb, prm, D, return_all=True, **kw)
    torch.cuda.synchronize()
    assert_same(host(got["volL"]), want["volL"], "left.bin volume")
    assert_same(host(got["volR"]), want["volR"], "right.bin volume")
    assert_same(host(got["dispL0"]), want["dispL0"], "left argmin")
    assert_same(host(got["dispR0"]), want["dispR0"], "right argmin")
    assert_same(host(got["disp"]), want["disp"], "disp.bin")


SWITCHES = [dict(sm_terminate=t) for t in ("cnn", "cbca1", "sgm", "cbca2", "occlusion", "mismatch", "subpixel_enchancement",
                                            "median", "bilateral")] + \
           [dict(sm_skip=k) for k in ("cbca", "sgm", "occlusion", "subpixel_enchancement", "median", "bilateral")]


@pytest.mark.parametrize("sw", SWITCHES, ids=lambda d: "%s=%s" % tuple(d.items())[0])
@pytest.mark.parametrize("name,C", [("kitti_fast", 16), ("kitti2015_slow", 0), ("mb_slow", 0)])
def test_sm_terminate_and_skip(mc, oracle, sw, name, C):
    """-sm_terminate <stage> / -sm_skip <stage> (main.lua:25-26, 956, 988-1079) in the fused entry point."""
    H, W, D = 20, 72, 20
    prm = dict(mc.PRESETS[name])
    if name == "mb_slow":
        prm["cbca_i2"] = 2
    prm.update(sw)
    x0, x1 = smooth_pair(H, W, 10, seed=31)
    xb = dev(np.stack([x0, x1]))[:, None]
    if C:
        f = features(C, H, W, seed=5)
        want = oracle.stereo_predict(prm, x0, x1, D, featL=f[0], featR=f[1])
        kw = dict(feat=dev(f))
    else:
        vl, vr = raw_volumes(D, H, W, seed=7)
        want = oracle.stereo_predict(prm, x0, x1, D, rawL=vl, rawR=vr)
        kw = dict(raw=(dev(vl), dev(vr)))
    got = mc.stereo_predict_fused(xb, prm, D, want_volumes=True, want_disp0=True, **kw)
    torch.cuda.synchronize()
    for k in ("volL", "volR", "dispL0", "dispR0", "disp"):
        assert_same(host(got[k]), want[k], k)


def test_main_predict_writes_the_reference_files(mc, oracle, tmp_path, monkeypatch):
    """`main.py kitti fast -a predict -left L.png -right R.png -disp_max D` (main.lua:1084-1105): left.bin / right.bin /
    disp.bin with the reference's layout; contents bit-exact vs the oracle fed with the same features."""
    from PIL import Image
    from mc_cnn_amd import main as mcmain
    H, W, D = 40, 96, 24
    rng = np.random.default_rng(3)
    from scipy.ndimage import gaussian_filter
    base = gaussian_filter(rng.random((H, W + 8)), 2.0)
    base = ((base - base.min()) / np.ptp(base) * 255).astype(np.uint8)
    Image.fromarray(base[:, 8:]).save(tmp_path / "L.png")
    Image.fromarray(np.stack([base[:, :W]] * 3, axis=-1)).save(tmp_path / "R.png")  # RGB on purpose: rgb2y path
    Image.fromarray(np.stack([base[:, 8:]] * 3, axis=-1)).save(tmp_path / "L3.png")
    monkeypatch.chdir(tmp_path)
    assert mcmain.main(["kitti", "fast", "-a", "predict", "-net_fname", "random:7", "-left", "L3.png", "-right", "R.png",
                        "-disp_max", str(D)]) == 0
    left = mc.read_bin("left.bin", (1, D, H, W))
    right = mc.read_bin("right.bin", (1, D, H, W))
    disp = mc.read_bin("disp.bin", (1, 1, H, W))
    assert (tmp_path / "left.bin").stat().st_size == 4 * D * H * W
    # the same host path by hand, then the oracle on the resulting features
    x0 = mcmain.normalize(mcmain.rgb2y(mcmain.load_image("L3.png")))
    x1 = mcmain.normalize(mcmain.rgb2y(mcmain.load_image("R.png")))
    xb = dev(np.stack([x0, x1]))
    layers = mcmain.load_net("random:7", "kitti", "fast")
    feat = host(mcmain.features_fast(xb, layers))
    prm = dict(mc.TABLES[("kitti", "fast")])
    prm["border_n"] = len(layers)
    want = oracle.stereo_predict(prm, x0[0], x1[0], D, featL=feat[0], featR=feat[1])
    assert_same(left, want["volL"], "left.bin")
    assert_same(right, want["volR"], "right.bin")
    assert_same(disp, want["disp"], "disp.bin")


@pytest.mark.parametrize("arch", ["ad", "census"])
def test_main_predict_hand_crafted_costs(mc, oracle, tmp_path, monkeypatch, arch):
    """`main.py kitti ad|census -a predict` (main.lua:932-942): cost volumes straight from the image pair."""
    from PIL import Image
    from mc_cnn_amd import main as mcmain
    H, W, D = 33, 70, 12
    rng = np.random.default_rng(9)
    from scipy.ndimage import gaussian_filter
    base = gaussian_filter(rng.random((H, W + 5)), 1.5)
    base = ((base - base.min()) / np.ptp(base) * 255).astype(np.uint8)
    Image.fromarray(base[:, 5:]).save(tmp_path / "L.png")
    Image.fromarray(base[:, :W]).save(tmp_path / "R.png")
    monkeypatch.chdir(tmp_path)
    assert mcmain.main(["kitti", arch, "-a", "predict", "-left", "L.png", "-right", "R.png", "-disp_max", str(D)]) == 0
    x0 = mcmain.normalize(mcmain.load_image("L.png"))
    x1 = mcmain.normalize(mcmain.load_image("R.png"))
    prm = dict(mc.TABLES[("kitti", arch)])
    cost = oracle.ad if arch == "ad" else oracle.census
    rawL, rawR = cost(x0, x1, D, -1), cost(x1, x0, D, 1)
    want = oracle.stereo_predict(prm, x0[0], x1[0], D, rawL=rawL, rawR=rawR)
    assert_same(mc.read_bin("left.bin", (1, D, H, W)), want["volL"], "left.bin")
    assert_same(mc.read_bin("right.bin", (1, D, H, W)), want["volR"], "right.bin")
    assert_same(mc.read_bin("disp.bin", (1, 1, H, W)), want["disp"], "disp.bin")


def test_predict_kitti_end_to_end(mc, oracle, tmp_path, capsys):
    """`predict_kitti.py test` over two synthetic KITTI-layout pairs: the printed mean 3-pixel error is the one of the
    disparity maps the oracle produces for the same features."""
    from PIL import Image
    from scipy.ndimage import gaussian_filter
    from mc_cnn_amd import binio, main as mcmain, predict_kitti as pk
    root = tmp_path / "unzip"
    for sub in ("training/image_0", "training/image_1", "training/disp_noc"):
        (root / sub).mkdir(parents=True)
    H, W, D = 30, 80, 12
    rng = np.random.default_rng(4)
    want_errs = []
    layers = mcmain.load_net("random:7", "kitti", "fast")
    prm = dict(mc.TABLES[("kitti", "fast")])
    prm["border_n"] = len(layers)
    for i in range(2):
        base = gaussian_filter(rng.random((H, W + 6)), 2.0)
        base = ((base - base.min()) / np.ptp(base) * 255).astype(np.uint8)
        Image.fromarray(base[:, 6:]).save(root / ("training/image_0/%06d_10.png" % i))
        Image.fromarray(base[:, :W]).save(root / ("training/image_1/%06d_10.png" % i))
        gt = np.full((H, W), 6.0, np.float32)
        gt[:, :10] = 0
        binio.write_png16(gt, str(root / ("training/disp_noc/%06d_10.png" % i)))
        x0 = mcmain.normalize(mcmain.load_image(str(root / ("training/image_0/%06d_10.png" % i))))
        x1 = mcmain.normalize(mcmain.load_image(str(root / ("training/image_1/%06d_10.png" % i))))
        feat = host(mcmain.features_fast(dev(np.stack([x0, x1])), layers))
        disp = oracle.stereo_predict(prm, x0[0], x1[0], D, featL=feat[0], featR=feat[1])["disp"].reshape(H, W)
        want_errs.append(pk.three_pixel_error(disp, binio.read_png16(str(root / ("training/disp_noc/%06d_10.png" % i)))))
    for k in ("1", "2", "3"):   # pairs in flight per rank: the results do not depend on it
        assert pk.main(["test", "-path", str(root), "-net_fname", "random:7", "-disp_max", str(D), "-n", "2", "-pairs_in_flight", k]) == 0
        lines = capsys.readouterr().out.strip().splitlines()
        assert [float(l.split()[1]) for l in lines[:2]] == want_errs
        assert abs(float(lines[-1]) - sum(want_errs) / 2) < 1e-12


def test_errors_are_loud(mc):
    """Bad arguments raise (reference: Lua error), they never fall back."""
    t = torch.zeros((1, 4, 8, 8), device="cuda")
    with pytest.raises(mc._lib.McError):
        mc.adcensus.median2d(t[:, :1], 4)       # even kernel size (adcensus.cu:1601)
    with pytest.raises(mc._lib.McError):
        mc.adcensus.cbca(t, t, t, t, -1)        # in-place
    with pytest.raises(TypeError):
        mc.adcensus.cross(torch.zeros(8, 8), t, 1, 0.1)  # CPU tensor
