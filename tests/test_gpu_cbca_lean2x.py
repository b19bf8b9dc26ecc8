"""-m gpu: TWO cross-based aggregation passes in one launch (cbca_lean2x_kernel; hook forms 10 / 11), the form mc_predict runs pairs
of passes in on textured pairs: a wave computes the first pass's rows it needs into an LDS tile, its own outputs of the second pass
out of that tile, and redoes the listed outputs (supports that are not the minimal 3 x 3) of both passes with the reference's loop
-- against the oracle's cbca applied twice, bit for bit: textures (few listed outputs), real-scene and blocky arms (most outputs
listed, supports reaching over the tile's edge: first-pass values recomputed on the spot), images smaller than a tile, ragged widths,
widths around the strip pitch of 252, both directions, every rows-per-wave instance, special values; a list that does not fit / is
another problem's (the strip kernel takes both passes); mc_predict with even and odd pass counts."""
import numpy as np
import pytest

from util import blocky_pair, diff_report, natural_pair, random_pair, raw_volumes, same_bits, smooth_pair

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def pair(mk, H, W, D):
    return {"smooth": lambda: smooth_pair(H, W, 8, seed=H), "random": lambda: random_pair(H, W, seed=W),
            "blocky": lambda: blocky_pair(H, W, seed=D), "natural": lambda: natural_pair(H, W, 8, seed=H + W, sigma=8.0),
            "flat": lambda: (np.zeros((H, W), np.float32), np.zeros((H, W), np.float32))}[mk]()


def twice(oracle, x0c, x1c, vol, direction):
    with np.errstate(all="ignore"):
        return oracle.cbca(x0c, x1c, oracle.cbca(x0c, x1c, vol, direction), direction)


SHAPES = [(90, 300, 9), (41, 519, 6), (27, 253, 5), (83, 64, 12), (37, 449, 4), (140, 130, 3), (5, 7, 3), (16, 256, 8), (17, 257, 9),
          (3, 1030, 5), (1, 9, 2), (9, 1, 2), (19, 252, 3), (20, 251, 3), (21, 254, 3), (33, 504, 2), (8, 506, 2), (9, 758, 2)]


@pytest.mark.parametrize("H,W,D", SHAPES)
@pytest.mark.parametrize("mk,L1,tau1", [("smooth", 14, 0.02), ("smooth", 9, 0.2), ("random", 14, 0.5), ("natural", 14, 0.02),
                                        ("blocky", 14, 0.2), ("blocky", 34, 10.0), ("smooth", 0, 0.0)])
def test_two_passes_in_one_launch(mc, oracle, H, W, D, mk, L1, tau1):
    x0, x1 = pair(mk, H, W, D)
    x0c, x1c = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=13)
    v2l, v2r = raw_volumes(D, H, W, seed=14)
    for direction, vol, vol2 in ((-1, vl, v2l), (1, vr, v2r)):
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol), out, direction, form=10)
        got, want = out.cpu().numpy(), twice(oracle, x0c, x1c, vol, direction)
        assert same_bits(got, want), diff_report(got, want, "listing launch, dir=%d" % direction)
        hdr = mc.adcensus.cbca_cfg_list_header(out.device, D, H, W, 10)
        assert hdr[2:5] == [D, H, W] and hdr[5] == direction + 1 and hdr[7] == 0x108, hdr
        if mk == "smooth" and L1 == 14 and H >= 16 and W >= 64:   # a texture: the two-pass kernel itself must have run
            assert hdr[1] == 0, "the texture's list was declared unusable: %r" % (hdr,)
        out = torch.full((1, D, H, W), -7.0, device="cuda")   # another volume of the same pair out of the same list
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol2), out, direction, form=11)
        got, want = out.cpu().numpy(), twice(oracle, x0c, x1c, vol2, direction)
        assert same_bits(got, want), diff_report(got, want, "launch reading the list, dir=%d" % direction)


@pytest.mark.parametrize("H,W,D", [(61, 530, 5), (90, 300, 9), (5, 7, 3), (17, 257, 9), (3, 1030, 5), (140, 130, 3), (1, 9, 2), (25, 760, 3)])
@pytest.mark.parametrize("rb", [0, 4, 8, 12, 5])   # rows per wave (anything but 4 / 8 / 12: the product's choice)
@pytest.mark.parametrize("mk,L1,tau1", [("smooth", 14, 0.05), ("blocky", 14, 0.2), ("natural", 14, 0.02)])
def test_two_pass_wave_geometries(mc, oracle, H, W, D, rb, mk, L1, tau1):
    x0, x1 = pair(mk, H, W, D)
    x0c, x1c = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=5)
    v2l, v2r = raw_volumes(D, H, W, seed=6)
    for direction, vol, vol2 in ((-1, vl, v2l), (1, vr, v2r)):
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol), out, direction, rb=rb, d0=1 << 20, form=10)   # (d0: no cost limit -- the kernel runs wherever the list fits)
        got, want = out.cpu().numpy(), twice(oracle, x0c, x1c, vol, direction)
        assert same_bits(got, want), diff_report(got, want, "rb=%d dir=%d" % (rb, direction))
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol2), out, direction, rb=rb, form=11)
        got, want = out.cpu().numpy(), twice(oracle, x0c, x1c, vol2, direction)
        assert same_bits(got, want), diff_report(got, want, "reading the list: rb=%d dir=%d" % (rb, direction))


def test_two_pass_special_values(mc, oracle):
    """zeros, negative zeros, denormals, huge values, infinities and NaNs inside the valid region, through both passes"""
    H, W, D = 40, 260, 6
    x0, x1 = smooth_pair(H, W, 8, seed=8)
    x0c, x1c = oracle.cross(x0, 14, 0.1), oracle.cross(x1, 14, 0.1)
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
    want = twice(oracle, x0c, x1c, vl, -1)
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, form=10)
    got = out.cpu().numpy()
    assert same_bits(got, want), diff_report(got, want, "two passes, special values")


def test_two_pass_falls_back_to_the_strip_kernel(mc, oracle):
    """(a) a list written for another problem (other direction, or the single-pass wave geometry) is not used; (b) a list that cannot
    hold the pair's entries (the hook's nd = capacity in 16-byte slots) -- the strip kernel takes both passes, results stay exact"""
    H, W, D = 33, 140, 4
    x0, x1 = smooth_pair(H, W, 8, seed=2)
    x0c, x1c = oracle.cross(x0, 14, 0.05), oracle.cross(x1, 14, 0.05)
    vl, vr = raw_volumes(D, H, W, seed=5)
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, form=10)
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vr), out, 1, form=11)     # the list on the scratch is direction -1's
    got, want = out.cpu().numpy(), twice(oracle, x0c, x1c, vr, 1)
    assert same_bits(got, want), diff_report(got, want, "list of the other direction")
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, rb=8, form=8)   # the single-pass kernel's list (8 rows per wave)
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, rb=8, form=11)
    got, want = out.cpu().numpy(), twice(oracle, x0c, x1c, vl, -1)
    assert same_bits(got, want), diff_report(got, want, "list of the single-pass geometry")
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, rb=4, form=10)
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, rb=8, form=11)   # written for 4 rows per wave
    got = out.cpu().numpy()
    assert same_bits(got, want), diff_report(got, want, "list of another wave geometry")
    for Hf, Wf, Df in ((64, 300, 2), (200, 600, 1)):
        z = np.zeros((Hf, Wf), np.float32)
        zc = oracle.cross(z, 14, 1.0)
        v, _ = raw_volumes(Df, Hf, Wf, seed=6)
        want = twice(oracle, zc, zc, v, -1)
        for form in (10, 11):
            out = torch.full((1, Df, Hf, Wf), -7.0, device="cuda")
            mc.adcensus.cbca_cfg(dev(zc), dev(zc), dev(v), out, -1, nd=1000, form=form)
            got = out.cpu().numpy()
            assert same_bits(got, want), diff_report(got, want, "every output listed, form %d" % form)


@pytest.mark.parametrize("i1,i2", [(2, 3), (2, 4), (1, 2), (3, 0), (0, 5)])
def test_fused_predict_on_a_texture_runs_pairs_of_passes(mc, oracle, i1, i2):
    """mc_predict with the Middlebury parameter set (L1 = 14) on a textured pair: passes go in pairs through cbca_lean2x, an odd last
    one through the strip kernel -- all five outputs against the oracle"""
    H, W, D = 70, 420, 24
    prm = dict(mc.PRESETS["mb_slow"], cbca_i1=i1, cbca_i2=i2)
    x0, x1 = smooth_pair(H, W, 10, seed=9)
    vl, vr = raw_volumes(D, H, W, seed=5)
    want = oracle.stereo_predict(prm, x0, x1, D, rawL=vl, rawR=vr)
    xb = dev(np.stack([x0, x1]))[:, None]
    got = mc.stereo_predict_fused(xb, prm, D, raw=(dev(vl), dev(vr)), want_volumes=True, want_disp0=True)
    for k in ("volL", "volR", "dispL0", "dispR0", "disp"):
        g = got[k].cpu().numpy()
        assert same_bits(g, want[k]), diff_report(g, want[k], k)


def test_fused_predict_on_real_scene_arms_is_unchanged(mc, oracle):
    """... and on a pair whose route is the tile kernel's, the pairs of passes go through the SGM's scratch volume"""
    H, W, D = 60, 300, 16
    prm = dict(mc.PRESETS["mb_slow"], cbca_i2=3)
    x0, x1 = natural_pair(H, W, 8, seed=4, sigma=8.0)
    vl, vr = raw_volumes(D, H, W, seed=7)
    want = oracle.stereo_predict(prm, x0, x1, D, rawL=vl, rawR=vr)
    xb = dev(np.stack([x0, x1]))[:, None]
    got = mc.stereo_predict_fused(xb, prm, D, raw=(dev(vl), dev(vr)), want_volumes=True, want_disp0=True)
    for k in ("volL", "volR", "dispL0", "dispR0", "disp"):
        g = got[k].cpu().numpy()
        assert same_bits(g, want[k]), diff_report(g, want[k], k)
