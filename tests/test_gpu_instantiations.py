"""-m gpu: the kernel instantiations that only the benchmarked SIZES select, forced at small shapes through the
test hooks of the C ABI (mc_cbca_ws_cfg, mc_transpose_cfg) and compared with the oracle:

  * rows per CBCA strip rb in {16, 25, 40} (KITTI size runs 25, 1000x1500 runs 40; small shapes pick 16),
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
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol), out, direction, rb=rb, nt=nt)
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
@pytest.mark.parametrize("R,Cn", [(228, 370 * 7), (65, 129), (256, 1000)])
def test_transposes_forced_cache_policy(mc, R, Cn, nt):
    rng = np.random.default_rng(R)
    a = rng.standard_normal((R, Cn)).astype(np.float32)
    a[0, :5] = np.nan
    out = torch.empty((Cn, R), device="cuda")
    mc.adcensus.transpose_cfg(dev(a), out, R, Cn, Cn, R, scale=0.25, nt=nt)
    assert same_bits(out.cpu().numpy(), (a * np.float32(0.25)).T)


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
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, rb=rb, nt=nt)
        got = out.cpu().numpy()
        assert same_bits(got, want), diff_report(got, want, "special values rb=%d nt=%d" % (rb, nt))


# ---- the window kernel (cbca form 2: what mc_predict takes for L1 <= 5) ---------------------------------------------
@pytest.mark.parametrize("H,W,D", [(90, 300, 9), (41, 519, 6), (27, 253, 5), (83, 64, 12), (37, 249, 4), (12, 497, 3)])
@pytest.mark.parametrize("rb", [0, 16, 40])
@pytest.mark.parametrize("mk,L1,tau1", [("smooth", 5, 0.13), ("random", 5, 0.5), ("blocky", 5, 0.2), ("natural", 5, 0.13),
                                        ("natural", 3, 0.03), ("blocky", 2, 0.3), ("smooth", 0, 0.0), ("natural", 4, 1.0)])
def test_cbca_window_kernel(mc, oracle, H, W, D, rb, mk, L1, tau1):
    """short arms (<= 4): every kind of image, strips with ragged edges, all row-chunk sizes, both cache policies, both
    directions -- against the oracle, bit for bit"""
    from util import natural_pair
    x0, x1 = {"smooth": lambda: smooth_pair(H, W, 8, seed=H), "random": lambda: random_pair(H, W, seed=W),
              "blocky": lambda: blocky_pair(H, W, seed=D), "natural": lambda: natural_pair(H, W, 8, seed=H + W, sigma=8.0)}[mk]()
    x0c, x1c = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=13)
    for direction, vol in ((-1, vl), (1, vr)):
        want = oracle.cbca(x0c, x1c, vol, direction)
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol), out, direction, rb=rb, nt=(H + rb) & 1, form=2)
        got = out.cpu().numpy()
        assert same_bits(got, want), diff_report(got, want, "window kernel rb=%d dir=%d" % (rb, direction))


def test_cbca_window_kernel_special_values(mc, oracle):
    """zeros, negative zeros, denormals, huge values, infinities and NaNs inside the valid region: a tap that is not in the
    support is never an operand (an inf / NaN next to a support must not leak into it), a support of nothing but -0.0
    sums to +0.0"""
    H, W, D = 40, 260, 6
    x0, x1 = blocky_pair(H, W, seed=8)
    x0c, x1c = oracle.cross(x0, 5, 0.2), oracle.cross(x1, 5, 0.2)
    vl, _ = raw_volumes(D, H, W, seed=3)
    rng = np.random.default_rng(1)
    vl[0, :, 20:] = 0.0
    vl[1, :, 20:] = rng.random((H, W - 20)).astype(np.float32) * np.float32(1e-42)
    vl[2, :, 20:] = rng.random((H, W - 20)).astype(np.float32) * np.float32(1e-30)
    vl[3, :, 20:] = rng.random((H, W - 20)).astype(np.float32) * np.float32(3e38)
    for k in range(40):
        vl[4, rng.integers(0, H), rng.integers(20, W)] = np.inf if k & 1 else np.nan
    vl[5, :, 20:] = -rng.random((H, W - 20)).astype(np.float32) * np.float32(1e-38)
    vl[5, 25:, 20:] = -0.0
    with np.errstate(all="ignore"):
        want = oracle.cbca(x0c, x1c, vl, -1)
    for rb, nt in ((0, -1), (25, 1)):
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, rb=rb, nt=nt, form=2)
        got = out.cpu().numpy()
        assert same_bits(got, want), diff_report(got, want, "window kernel special values rb=%d nt=%d" % (rb, nt))


def test_cbca_forms_agree_on_a_realistic_pair(mc):
    """strip kernel, window kernel and the one-thread-per-voxel kernel on a pair with real-scene arm statistics"""
    from util import natural_pair
    H, W, D = 120, 700, 20
    x0, x1 = natural_pair(H, W, D, seed=5)
    xb = dev(np.stack([x0, x1]))[:, None]
    x0c = torch.empty((1, 4, H, W), device="cuda"); x1c = torch.empty_like(x0c)
    mc.adcensus.cross(xb[0:1], x0c, 5, 0.13); mc.adcensus.cross(xb[1:2], x1c, 5, 0.13)
    vin = torch.rand((1, D, H, W), device="cuda")
    outs = []
    for form in (1, 2):
        o = torch.full_like(vin, -7.0)
        mc.adcensus.cbca_cfg(x0c, x1c, vin, o, -1, form=form)
        outs.append(o.cpu().numpy())
    o = torch.full_like(vin, -7.0)
    mc.adcensus.cbca_reference_shaped(x0c, x1c, vin, o, -1)
    assert same_bits(outs[0], o.cpu().numpy()) and same_bits(outs[1], o.cpu().numpy())


# ---- strip kernel + the pair's list of large supports (cbca form 3: what mc_predict takes for L1 > 5) -----------------
@pytest.mark.parametrize("H,W,D", [(90, 300, 9), (41, 519, 6), (60, 253, 5), (83, 64, 12), (37, 449, 4), (140, 230, 3)])
@pytest.mark.parametrize("rb", [0, 25])
@pytest.mark.parametrize("mk,L1,tau1", [("smooth", 14, 0.02), ("natural", 14, 0.02), ("blocky", 14, 0.2), ("natural", 9, 0.05),
                                        ("blocky", 6, 0.3), ("natural", 5, 0.13), ("random", 14, 2.5), ("blocky", 34, 10.0)])
def test_cbca_listed(mc, oracle, H, W, D, rb, mk, L1, tau1):
    """supports that do not fit the strip kernel's window form come from the list kernel: every kind of arm statistics, up
    to arms as long as the image allows (("blocky", 34, 10.0)), both directions, both cache policies"""
    from util import natural_pair
    x0, x1 = {"smooth": lambda: smooth_pair(H, W, 8, seed=H), "random": lambda: random_pair(H, W, seed=W),
              "blocky": lambda: blocky_pair(H, W, seed=D), "natural": lambda: natural_pair(H, W, 8, seed=H + W, sigma=8.0)}[mk]()
    x0c, x1c = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=13)
    for direction, vol in ((-1, vl), (1, vr)):
        want = oracle.cbca(x0c, x1c, vol, direction)
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol), out, direction, rb=rb, nt=(H + rb) & 1, form=3)
        got = out.cpu().numpy()
        assert same_bits(got, want), diff_report(got, want, "listed rb=%d dir=%d" % (rb, direction))


def test_cbca_listed_special_values(mc, oracle):
    H, W, D = 40, 260, 6
    x0, x1 = blocky_pair(H, W, seed=8)
    x0c, x1c = oracle.cross(x0, 14, 0.2), oracle.cross(x1, 14, 0.2)
    vl, _ = raw_volumes(D, H, W, seed=3)
    rng = np.random.default_rng(1)
    vl[0, :, 20:] = 0.0
    vl[1, :, 20:] = rng.random((H, W - 20)).astype(np.float32) * np.float32(1e-42)
    vl[3, :, 20:] = rng.random((H, W - 20)).astype(np.float32) * np.float32(3e38)
    for k in range(40):
        vl[4, rng.integers(0, H), rng.integers(20, W)] = np.inf if k & 1 else np.nan
    vl[5, 25:, 20:] = -0.0
    with np.errstate(all="ignore"):
        want = oracle.cbca(x0c, x1c, vl, -1)
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, form=3)
    got = out.cpu().numpy()
    assert same_bits(got, want), diff_report(got, want, "listed, special values")
