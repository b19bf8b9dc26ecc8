"""-m gpu: the kernel instantiations that only the benchmarked SIZES select, forced at small shapes through the
test hooks of the C ABI (mc_cbca_ws_cfg, mc_transpose_cfg) and compared with the oracle:

  * rows per CBCA strip rb in {16, 25, 40} of the strip kernel (KITTI size runs 25, 1000x1500 runs 40; small shapes pick 16),
  * the route adcensus.cbca takes by the pair's arms (tile kernel short / long arms, strip kernel, one thread per voxel),
  * the non-temporal instantiations of the CBCA strip kernel and of the layout transposes (selected above 768 MB),
  * plane sub-ranges of a volume.
(tests/test_gpu_fullsize.py checks the same code at the real sizes against the reference's kernels.)"""
import numpy as np
import pytest

from util import blocky_pair, diff_report, random_pair, raw_volumes, same_bits, smooth_pair

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


@pytest.mark.parametrize("H,W,D", [(90, 300, 9), (41, 519, 6), (27, 253, 5), (83, 64, 12)])
@pytest.mark.parametrize("rb", [16, 25, 40])
@pytest.mark.parametrize("nt", [0, 1])
@pytest.mark.parametrize("mk,L1,tau1", [("smooth", 14, 0.02), ("random", 5, 0.13), ("blocky", 14, 0.2)])
def test_cbca_forced_rows_and_cache_policy(mc, oracle, H, W, D, rb, nt, mk, L1, tau1):
    x0, x1 = {"smooth": lambda: smooth_pair(H, W, 8, seed=H), "random": lambda: random_pair(H, W, seed=W),
              "blocky": lambda: blocky_pair(H, W, seed=D)}[mk]()
    x0c, x1c = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=13)
    for direction, vol in ((-1, vl), (1, vr)):
        want = oracle.cbca(x0c, x1c, vol, direction)
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol), out, direction, rb=rb, nt=nt, form=1)
        got = out.cpu().numpy()
        assert same_bits(got, want), diff_report(got, want, "cbca rb=%d nt=%d dir=%d" % (rb, nt, direction))


def test_cbca_plane_range(mc, oracle):
    H, W, D = 30, 100, 11
    x0, x1 = smooth_pair(H, W, 8, seed=4)
    x0c, x1c = oracle.cross(x0, 14, 0.05), oracle.cross(x1, 14, 0.05)
    vl, _ = raw_volumes(D, H, W, seed=2)
    want = oracle.cbca(x0c, x1c, vl, -1)
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, d0=3, nd=5)
    got = out.cpu().numpy()[0]
    assert same_bits(got[3:8], want[3:8]), diff_report(got[3:8], want[3:8], "planes 3..7")
    assert (got[:3] == -7.0).all() and (got[8:] == -7.0).all(), "planes outside [d0, d0+nd) were written"


@pytest.mark.parametrize("nt", [0, 1])
@pytest.mark.parametrize("R,Cn", [(228, 370 * 7), (65, 129), (256, 1000), (228, 2592), (4, 8), (68, 132), (228, 453620 // 19),
                                  (248, 640), (72, 64), (100, 4100), (12, 256), (252, 128)])   # (16-byte and 4-byte paths; round 5: whole-run kernel for short rows off the 128-byte grid)
def test_transposes_forced_cache_policy(mc, R, Cn, nt):
    rng = np.random.default_rng(R)
    a = rng.standard_normal((R, Cn)).astype(np.float32)
    a[0, :5] = np.nan
    out = torch.empty((Cn, R), device="cuda")
    mc.adcensus.transpose_cfg(dev(a), out, R, Cn, Cn, R, scale=0.25, nt=nt)
    assert same_bits(out.cpu().numpy(), (a * np.float32(0.25)).T)


@pytest.mark.parametrize("nt", [0, 1])
@pytest.mark.parametrize("R,Cn,ld", [(70, 1300, 72), (229, 260, 232), (228, 4000, 228), (13, 64, 16), (228, 64, 256), (9, 68, 12), (8, 4096, 8)])
def test_transpose_whole_runs_with_padded_rows(mc, R, Cn, ld, nt):
    """(D,H,W) -> (H,W,ds) as mc_predict does it: output rows of ds = D rounded up to 4 floats, the padding never written"""
    rng = np.random.default_rng(R + Cn)
    a = rng.standard_normal((R, Cn)).astype(np.float32)
    a[R // 2, ::7] = np.nan
    out = torch.full((Cn, ld), -7.0, device="cuda")
    mc.adcensus.transpose_cfg(dev(a), out, R, Cn, Cn, ld, scale=1.0, nt=nt)
    got = out.cpu().numpy()
    assert same_bits(got[:, :R], a.T)
    assert (got[:, R:] == -7.0).all(), "the padding behind a run was written"


def test_left_only_skips_the_right_volume_without_changing_the_left(mc):
    """mc_params.left_only: dataset mb outside `-a predict` runs direction -1 only (main.lua:953-955); disp is the same"""
    H, W, D = 40, 120, 24
    prm = dict(mc.PRESETS["mb_slow"], cbca_i2=3)
    x0, x1 = smooth_pair(H, W, 10, seed=9)
    xb = dev(np.stack([x0, x1]))[:, None]
    vl, vr = raw_volumes(D, H, W, seed=5)
    raw = (dev(vl), dev(vr))
    both = mc.stereo_predict_fused(xb, prm, D, raw=raw)["disp"].cpu().numpy()
    left = mc.stereo_predict_fused(xb, dict(prm, left_only=1), D, raw=raw)["disp"].cpu().numpy()
    assert same_bits(left, both)
    # a right-side output asked for: both directions run regardless
    r = mc.stereo_predict_fused(xb, dict(prm, left_only=1), D, raw=raw, want_disp0=True)
    w = mc.stereo_predict_fused(xb, prm, D, raw=raw, want_disp0=True)
    assert same_bits(r["dispR0"].cpu().numpy(), w["dispR0"].cpu().numpy())
    # kitti (LR check): the flag is ignored
    prk = dict(mc.PRESETS["kitti_slow"])
    a = mc.stereo_predict_fused(xb, prk, D, raw=raw)["disp"].cpu().numpy()
    b = mc.stereo_predict_fused(xb, dict(prk, left_only=1), D, raw=raw)["disp"].cpu().numpy()
    assert same_bits(a, b)


def test_cbca_special_values(mc, oracle):
    """zeros, negative zeros, denormals, huge values, infinities and NaNs INSIDE the valid region, on a mix of minimal,
    window-form and larger supports: NaN / inf propagate as in the reference, and a support of nothing but -0.0 sums to
    +0.0 (the reference's accumulator starts at +0.0, adcensus.cu:356)."""
    H, W, D = 40, 260, 6
    x0, x1 = random_pair(H, W, seed=8)
    x0c, x1c = oracle.cross(x0, 14, 0.35), oracle.cross(x1, 14, 0.35)
    vl, _ = raw_volumes(D, H, W, seed=3)
    rng = np.random.default_rng(1)
    vl[0, :, 20:] = 0.0
    vl[1, :, 20:] = rng.random((H, W - 20)).astype(np.float32) * np.float32(1e-42)      # denormals
    vl[2, :, 20:] = rng.random((H, W - 20)).astype(np.float32) * np.float32(1e-30)      # tiny normals
    vl[3, :, 20:] = rng.random((H, W - 20)).astype(np.float32) * np.float32(3e38)       # sums overflow
    vl[4, 10, 100] = np.inf
    vl[4, 20, 150] = np.nan
    vl[5, :, 20:] = -rng.random((H, W - 20)).astype(np.float32) * np.float32(1e-38)
    vl[5, 25:, 20:] = -0.0
    with np.errstate(all="ignore"):
        want = oracle.cbca(x0c, x1c, vl, -1)
    for rb, nt in ((0, -1), (25, 1)):
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, rb=rb, nt=nt, form=1)
        got = out.cpu().numpy()
        assert same_bits(got, want), diff_report(got, want, "special values rb=%d nt=%d" % (rb, nt))


# ---- adcensus.cbca's routes: the kernel the pair's arms call for is picked on the device (cbca_pack's route word) -------
@pytest.mark.parametrize("mk,L1,tau1,what", [
    ("smooth", 14, 0.02, "textured pair, nearly every support 3x3: strip kernel"),
    ("natural", 14, 0.02, "real-scene statistics, arms <= 13: tile kernel, long-arm instance"),
    ("natural", 5, 0.13, "arms <= 4: tile kernel, short-arm instance"),
    ("blocky", 34, 10.0, "arms as long as the image allows: strip kernel"),
    ("flat", 400, 1.0, "an arm longer than 254 pixels: one thread per voxel"),
])
@pytest.mark.parametrize("H,W,D", [(60, 300, 7), (37, 449, 4)])
def test_cbca_routes(mc, oracle, mk, L1, tau1, what, H, W, D):
    from util import natural_pair
    x0, x1 = {"smooth": lambda: smooth_pair(H, W, 8, seed=H), "blocky": lambda: blocky_pair(H, W, seed=D),
              "natural": lambda: natural_pair(H, W, 8, seed=H + W, sigma=8.0),
              "flat": lambda: (np.zeros((H, W), np.float32), np.zeros((H, W), np.float32))}[mk]()
    x0c, x1c = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=13)
    for direction, vol in ((-1, vl), (1, vr)):
        want = oracle.cbca(x0c, x1c, vol, direction)
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca(dev(x0c), dev(x1c), dev(vol), out, direction)
        got = out.cpu().numpy()
        assert same_bits(got, want), diff_report(got, want, "adcensus.cbca (%s) dir=%d" % (what, direction))


def test_cbca_forms_agree_on_a_realistic_pair(mc):
    """strip kernel, both tile instances, adcensus.cbca's own choice and the one-thread-per-voxel kernel on a pair with
    real-scene arm statistics"""
    from util import natural_pair
    H, W, D = 120, 700, 20
    x0, x1 = natural_pair(H, W, D, seed=5)
    xb = dev(np.stack([x0, x1]))[:, None]
    for L1, tau1, forms in ((5, 0.13, (0, 1, 2, 3)), (14, 0.02, (0, 1, 3))):
        x0c = torch.empty((1, 4, H, W), device="cuda"); x1c = torch.empty_like(x0c)
        mc.adcensus.cross(xb[0:1], x0c, L1, tau1); mc.adcensus.cross(xb[1:2], x1c, L1, tau1)
        vin = torch.rand((1, D, H, W), device="cuda")
        ref = torch.full_like(vin, -7.0)
        mc.adcensus.cbca_reference_shaped(x0c, x1c, vin, ref, -1)
        for form in forms:
            o = torch.full_like(vin, -7.0)
            mc.adcensus.cbca_cfg(x0c, x1c, vin, o, -1, form=form)
            assert same_bits(o.cpu().numpy(), ref.cpu().numpy()), "L1=%d form %d" % (L1, form)
