"""-m gpu: the REFERENCE ITSELF as the checker.  oracle/_ref/libadcensus_ref.so is
/root/reference/adcensus.cu, unmodified, compiled for gfx950 (oracle/build_ref.py); its binding
functions are called by name exactly as main.lua calls them.  Each test checks, on the same inputs,
  (1) the CPU oracle (oracle/mc_oracle.c) against the reference  -> pins the oracle, and
  (2) this repository's HIP path (C ABI) against the reference    -> direct parity.
Bar: bit-exact including NaN masks."""
import numpy as np
import pytest

from util import (blocky_pair, diff_report, features, random_pair, raw_volumes, same_bits, smooth_pair)

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def same(got, want, name):
    assert same_bits(got, want), diff_report(got, want, name)


def test_registry_is_the_references(ref):
    """funcs[] (adcensus.cu:2061-2096): the 31 adcensus.* names + the two SpatialLogSoftMax ones."""
    names = ref.functions()
    assert len([n for n in names if n.startswith("adcensus.")]) == 31
    for n in ("StereoJoin", "ad", "census", "cross", "cbca", "sgm2", "spatial_argmin", "outlier_detection",
              "interpolate_occlusion", "interpolate_mismatch", "subpixel_enchancement", "median2d", "mean2d",
              "Normalize_forward"):
        assert "adcensus." + n in names


SHAPES = [(24, 40, 16), (17, 33, 7), (9, 70, 70), (12, 50, 64), (20, 300, 228)]


@pytest.mark.parametrize("H,W,D", SHAPES)
@pytest.mark.parametrize("C", [1, 64, 112])
def test_stereo_join(ref, mc, oracle, H, W, D, C):
    f = features(C, H, W, seed=H + W + C)
    fd = dev(f)
    rl = torch.full((1, D, H, W), float("nan"), device="cuda")
    rr = torch.full((1, D, H, W), float("nan"), device="cuda")
    ref.call("StereoJoin", fd[0:1].contiguous(), fd[1:2].contiguous(), rl, rr)
    ol, orr = oracle.stereo_join(f[0], f[1], D)
    same(ol, host(rl), "oracle volL vs reference")
    same(orr, host(rr), "oracle volR vs reference")
    vl = torch.full((1, D, H, W), float("nan"), device="cuda")
    vr = torch.full((1, D, H, W), float("nan"), device="cuda")
    mc.adcensus.StereoJoin(fd[0], fd[1], vl, vr)
    same(host(vl), host(rl), "hip volL vs reference")
    same(host(vr), host(rr), "hip volR vs reference")


@pytest.mark.parametrize("H,W,D", SHAPES[:4])
@pytest.mark.parametrize("direction", [-1, 1])
def test_ad_census(ref, mc, oracle, H, W, D, direction):
    x0, x1 = random_pair(H, W, seed=3)
    r = torch.empty((1, D, H, W), device="cuda")
    ref.call("ad", dev(x0)[None, None], dev(x1)[None, None], r, direction)
    same(oracle.ad(x0, x1, D, direction), host(r), "oracle ad vs reference")
    g = torch.empty((1, D, H, W), device="cuda")
    mc.adcensus.ad(dev(x0), dev(x1), g, direction)
    same(host(g), host(r), "hip ad vs reference")
    c0 = np.stack([x0, x1 * 0.5])
    c1 = np.stack([x1, x0 * 2.0])
    ref.call("census", dev(c0)[None], dev(c1)[None], r, direction)
    same(oracle.census(c0, c1, D, direction), host(r), "oracle census vs reference")
    mc.adcensus.census(dev(c0)[None], dev(c1)[None], g, direction)
    same(host(g), host(r), "hip census vs reference")


@pytest.mark.parametrize("H,W", [(24, 40), (17, 33), (40, 9), (1, 50), (30, 1), (64, 200)])
@pytest.mark.parametrize("L1,tau1", [(0, 0.0), (5, 0.13), (14, 0.02), (14, 1e9), (3, 0.5)])
def test_cross(ref, mc, oracle, H, W, L1, tau1):
    for mk in (random_pair, blocky_pair, lambda h, w, seed: smooth_pair(h, w, 8, seed=seed)):
        img, _ = mk(H, W, seed=5)
        r = torch.empty((1, 4, H, W), device="cuda")
        ref.call("cross", dev(img)[None], r, L1, tau1)
        same(oracle.cross(img, L1, tau1), host(r), "oracle cross vs reference")
        g = torch.empty((1, 4, H, W), device="cuda")
        mc.adcensus.cross(dev(img), g, L1, tau1)
        same(host(g), host(r), "hip cross vs reference")


@pytest.mark.parametrize("H,W,D", SHAPES[:4] + [(40, 120, 48)])
@pytest.mark.parametrize("L1,tau1", [(0, 0.0), (5, 0.4), (14, 1e9), (14, 0.02)])
@pytest.mark.parametrize("direction", [-1, 1])
def test_cbca(ref, mc, oracle, H, W, D, L1, tau1, direction):
    x0, x1 = blocky_pair(H, W, seed=11) if tau1 > 0.1 else smooth_pair(H, W, min(D, 8), seed=11)
    x0c, x1c = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=13)
    vol = vl if direction == -1 else vr
    r = torch.empty((1, D, H, W), device="cuda")
    ref.call("cbca", dev(x0c)[None], dev(x1c)[None], dev(vol)[None], r, direction)
    same(oracle.cbca(x0c, x1c, vol, direction), host(r), "oracle cbca vs reference")
    g = torch.empty((1, D, H, W), device="cuda")
    mc.adcensus.cbca(dev(x0c), dev(x1c), dev(vol), g, direction)
    same(host(g), host(r), "hip cbca vs reference")


SGM_PARAMS = [(4.0, 55.72, 0.02, 1.5, 3.0, 2.5), (1.32, 24.25, 0.08, 2.0, 3.0, 2.0), (1.3, 13.9, 0.13, 2.75, 4.5, 2.0)]


@pytest.mark.parametrize("H,W,D", SHAPES + [(6, 300, 260)])
@pytest.mark.parametrize("prm", SGM_PARAMS)
@pytest.mark.parametrize("direction", [-1, 1])
def test_sgm2(ref, mc, oracle, H, W, D, prm, direction):
    x0, x1 = smooth_pair(H, W, min(D, 8), seed=21)
    vl, vr = raw_volumes(D, H, W, seed=23)
    vol = oracle.dhw_to_hwd(vl if direction == -1 else vr)
    r = torch.zeros((1, H, W, D), device="cuda")
    tmp = torch.empty((W, D), device="cuda")
    ref.call("sgm2", dev(x0)[None], dev(x1)[None], dev(vol)[None], r, tmp, *prm, direction)
    same(oracle.sgm2(x0, x1, vol, *prm, direction), host(r), "oracle sgm2 vs reference")
    g = torch.zeros((1, H, W, D), device="cuda")
    mc.adcensus.sgm2(dev(x0), dev(x1), dev(vol)[None], g, None, *prm, direction)
    same(host(g), host(r), "hip sgm2 vs reference")


def test_sgm2_volume_of_two_gib(ref, mc):
    """A volume of 2 GiB or more takes the sweeps' 64-bit-address instances (sgm.hip: FAR), which no other shape reaches: sgm2 at
    560 x 3840 x 256 (2.2 GB) against the reference's kernels on the same device tensors, compared on the device."""
    H, W, D = 560, 3840, 256
    assert H * W * D * 4 >= 1 << 31
    g = torch.Generator(device="cuda").manual_seed(5)
    x0 = torch.rand((1, 1, H, W), device="cuda", generator=g)
    x1 = torch.rand((1, 1, H, W), device="cuda", generator=g)
    vol = torch.rand((1, H, W, D), device="cuda", generator=g)
    d = torch.arange(D, device="cuda")[None, None, None, :]
    xx = torch.arange(W, device="cuda")[None, None, :, None]
    vol.masked_fill_(d > xx, float("nan"))                    # the NaN triangle of a left volume
    prm = (1.3, 13.9, 0.13, 2.75, 4.5, 2.0)
    r = torch.zeros_like(vol)
    tmp = torch.empty((W, D), device="cuda")
    ref.call("sgm2", x0[0], x1[0], vol, r, tmp, *prm, -1)
    got = torch.zeros_like(vol)
    mc.adcensus.sgm2(x0[0, 0], x1[0, 0], vol, got, None, *prm, -1)
    torch.cuda.synchronize()
    same_dev = bool(((got.view(torch.int32) == r.view(torch.int32)) | (torch.isnan(got) & torch.isnan(r))).all().item())
    assert same_dev, "hip sgm2 (2.2 GB volume) differs from the reference's kernels"


@pytest.mark.parametrize("H,W,D", SHAPES[:4])
def test_spatial_argmin(ref, mc, oracle, H, W, D):
    vl, _ = raw_volumes(D, H, W, seed=31)
    vl[:, 0, 0] = np.nan
    vl[3 % D, 1, 1] = vl[:, 1, 1][~np.isnan(vl[:, 1, 1])].min()
    r = torch.empty((1, 1, H, W), device="cuda")
    ref.call("spatial_argmin", dev(vl)[None], r)
    same(oracle.argmin(vl) + 1, host(r), "oracle argmin(+1) vs reference spatial_argmin")
    g = torch.empty((1, 1, H, W), device="cuda")
    mc.adcensus.spatial_argmin(dev(vl)[None], g)
    same(host(g), host(r), "hip spatial_argmin vs reference")
    same(host(mc.adcensus.argmin(dev(vl)[None])) + 1, host(r), "hip argmin(+1) vs reference")


def _disp_maps(H, W, D, seed):
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    base = gaussian_filter(rng.random((H, W)), 4.0)
    base = (base - base.min()) / (np.ptp(base) + 1e-9) * (D - 1)
    d0 = np.floor(base).astype(np.float32)
    d1 = np.floor(np.roll(base, -3, axis=1)).astype(np.float32)
    noise = rng.random((H, W)) < 0.15
    d0[noise] = rng.integers(0, D, size=int(noise.sum())).astype(np.float32)
    return d0, d1


@pytest.mark.parametrize("H,W,D", [(24, 40, 16), (17, 33, 7), (40, 24, 12), (31, 130, 70), (60, 200, 100)])
def test_post_chain(ref, mc, oracle, H, W, D):
    d0, d1 = _disp_maps(H, W, D, seed=H + W)
    d0d, d1d = dev(d0)[None, None], dev(d1)[None, None]
    ro = torch.zeros((1, 1, H, W), device="cuda")
    ref.call("outlier_detection", d0d, d1d, ro, D)
    same(oracle.outlier_detection(d0, d1, D), host(ro), "oracle outlier vs reference")
    go = torch.zeros((1, 1, H, W), device="cuda")
    mc.adcensus.outlier_detection(d0d, d1d, go, D)
    same(host(go), host(ro), "hip outlier vs reference")

    rocc = ref.call("interpolate_occlusion", d0d, ro)[0]
    same(oracle.interpolate_occlusion(d0, host(ro)[0, 0]), host(rocc), "oracle occlusion vs reference")
    same(host(mc.adcensus.interpolate_occlusion(d0d, ro)), host(rocc), "hip occlusion vs reference")

    if (host(ro) != 2).any():  # all-mismatch input reads uninitialised memory in the reference (adcensus.cu:1054)
        rmis = ref.call("interpolate_mismatch", rocc, ro)[0]
        same(oracle.interpolate_mismatch(host(rocc)[0, 0], host(ro)[0, 0]), host(rmis), "oracle mismatch vs reference")
        same(host(mc.adcensus.interpolate_mismatch(rocc, ro)), host(rmis), "hip mismatch vs reference")
    else:
        rmis = rocc

    vl, _ = raw_volumes(D, H, W, seed=41)
    rsub = ref.call("subpixel_enchancement", rmis, dev(vl)[None], D)[0]
    same(oracle.subpixel_enchancement(host(rmis)[0, 0], vl), host(rsub), "oracle subpixel vs reference")
    same(host(mc.adcensus.subpixel_enchancement(rmis, dev(vl)[None], D)), host(rsub), "hip subpixel vs reference")

    for k in (1, 3, 5, 11):
        rmed = ref.call("median2d", rsub, k)[0]
        same(oracle.median2d(host(rsub)[0, 0], k), host(rmed), "oracle median%d vs reference" % k)
        same(host(mc.adcensus.median2d(rsub, k)), host(rmed), "hip median%d vs reference" % k)
    rmed = ref.call("median2d", rsub, 5)[0]
    from ref_pipeline import gaussian
    for sigma, t in ((1.67, 2.0), (5.99, 6.0), (7.74, 5.0)):
        kref = gaussian(sigma).float()
        same(oracle.gaussian(sigma), kref.numpy(), "oracle gaussian vs main.lua transliteration")
        same(mc.adcensus.gaussian(sigma).numpy(), kref.numpy(), "host gaussian vs main.lua transliteration")
        rmean = ref.call("mean2d", rmed, kref.cuda(), t)[0]
        same(oracle.mean2d(host(rmed)[0, 0], kref.numpy(), t), host(rmean), "oracle mean2d vs reference")
        same(host(mc.adcensus.mean2d(rmed, kref.cuda(), t)), host(rmean), "hip mean2d vs reference")


@pytest.mark.parametrize("C", [64, 1, 7, 112, 130])   # 64 / 112: the nets' widths (register-resident instances); 130: the any-width instance
def test_normalize_forward(ref, mc, oracle, C):
    rng = np.random.default_rng(9 + C)
    x = rng.standard_normal((2, C, 11, 23)).astype(np.float32)
    rn = torch.empty((2, 1, 11, 23), device="cuda")
    rout = torch.empty((2, C, 11, 23), device="cuda")
    ref.call("Normalize_forward", dev(x), rn, rout)
    same(oracle.normalize_forward(x), host(rout), "oracle normalize vs reference")
    gn = torch.empty_like(rn)
    gout = torch.empty_like(rout)
    mc.adcensus.Normalize_forward(dev(x), gn, gout)
    same(host(gout), host(rout), "hip normalize vs reference")
    same(host(gn), host(rn), "hip norm vs reference")


PRED_CASES = [
    ("kitti_fast", {}, 32, 96, 24, 64),
    ("kitti_fast", {}, 21, 70, 30, 16),
    ("kitti_fast", {"sgm_i": 2}, 16, 64, 16, 8),
    ("kitti_fast", {}, 40, 300, 228, 64),          # full KITTI disparity range on a row band
    ("kitti_slow", {}, 32, 96, 24, 0),
    ("kitti_slow", {"cbca_i2": 1}, 20, 60, 18, 0),
    ("mb_slow", {"cbca_i2": 3}, 28, 80, 20, 0),
    ("mb_slow", {}, 40, 120, 32, 0),               # the full 2+16 iterations
    ("mb_slow", {"cbca_i2": 2}, 28, 80, 20, 32),
]


@pytest.mark.parametrize("name,over,H,W,D,C", PRED_CASES)
def test_stereo_predict_vs_reference(ref, mc, oracle, name, over, H, W, D, C):
    """main.lua's stereo_predict over the reference's kernels vs the oracle and vs both HIP drivers:
    left.bin / right.bin contents, both arg-min maps and disp.bin, bit for bit."""
    from ref_pipeline import ref_stereo_predict
    prm = dict(mc.PRESETS[name])
    prm.update(over)
    x0, x1 = smooth_pair(H, W, min(D, 12), seed=77)
    xb = dev(np.stack([x0, x1]))[:, None]
    if C:
        f = features(C, H, W, seed=5)
        kw = dict(feat=dev(f))
        want_o = oracle.stereo_predict(prm, x0, x1, D, featL=f[0], featR=f[1])
    else:
        vl, vr = raw_volumes(D, H, W, seed=7)
        kw = dict(raw=(dev(vl), dev(vr)))
        want_o = oracle.stereo_predict(prm, x0, x1, D, rawL=vl, rawR=vr)
    want = ref_stereo_predict(ref, prm, xb, D, **kw)
    fused = mc.stereo_predict_fused(xb, prm, D, want_volumes=True, want_disp0=True, **kw)
    ops = mc.stereo_predict(xb, prm, D, return_all=True, **kw)
    torch.cuda.synchronize()
    for key, label in (("volL", "left.bin"), ("volR", "right.bin"), ("dispL0", "left argmin"),
                       ("dispR0", "right argmin"), ("disp", "disp.bin")):
        w = host(want[key])
        same(want_o[key], w, "oracle %s vs reference" % label)
        same(host(fused[key]), w, "hip fused %s vs reference" % label)
        same(host(ops[key]), w, "hip op-by-op %s vs reference" % label)
