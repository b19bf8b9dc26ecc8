"""-m gpu: cross-based aggregation of textured pairs as mc_predict runs it (cbca_lean.hip; hook forms 8 / 9): the pair's outputs
whose support is not the minimal 3 x 3 are listed once per direction (form 8), every pass then computes the minimal 3 x 3 mean
everywhere out of a register window and re-runs the reference's loop for the listed outputs (forms 8 and 9) -- against the
oracle, bit for bit: textures (few listed outputs), real-scene and blocky arms (most outputs listed), images smaller than a
strip, ragged widths (W not a multiple of 4: the row's last unit ends in the next row), both directions, both cache policies,
every wave geometry and launch variant, special values; a list that does not fit / is another problem's (strip kernel
takes over)."""
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


SHAPES = [(90, 300, 9), (41, 519, 6), (27, 253, 5), (83, 64, 12), (37, 449, 4), (140, 130, 3), (5, 7, 3), (16, 256, 8), (17, 257, 9),
          (3, 1030, 5), (1, 9, 2), (9, 1, 2)]


@pytest.mark.parametrize("H,W,D", SHAPES)
@pytest.mark.parametrize("mk,L1,tau1", [("smooth", 14, 0.02), ("smooth", 9, 0.2), ("random", 14, 0.5), ("natural", 14, 0.02),
                                        ("blocky", 14, 0.2), ("blocky", 34, 10.0), ("smooth", 0, 0.0)])
def test_lean_and_list(mc, oracle, H, W, D, mk, L1, tau1):
    x0, x1 = pair(mk, H, W, D)
    x0c, x1c = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=13)
    v2l, v2r = raw_volumes(D, H, W, seed=14)
    for direction, vol, vol2 in ((-1, vl, v2l), (1, vr, v2r)):
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol), out, direction, nt=(H + W) & 1, form=8)
        got, want = out.cpu().numpy(), oracle.cbca(x0c, x1c, vol, direction)
        assert same_bits(got, want), diff_report(got, want, "listing pass, dir=%d" % direction)
        out = torch.full((1, D, H, W), -7.0, device="cuda")   # another volume of the same pair out of the same list
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol2), out, direction, nt=H & 1, form=9)
        got, want = out.cpu().numpy(), oracle.cbca(x0c, x1c, vol2, direction)
        assert same_bits(got, want), diff_report(got, want, "pass reading the list, dir=%d" % direction)


@pytest.mark.parametrize("rb", [0, 2, 4, 8, 5])          # rows per wave (anything but 2 / 4 / 8: the product's choice)
@pytest.mark.parametrize("variant", [0, 4, 16, 20, 96])   # bit 2: the listed outputs in a launch of their own, bit 4: a band of rows per XCD, bits 5 / 6: non-temporal loads / stores
def test_lean_rows_per_wave_and_launch_variants(mc, oracle, rb, variant):
    H, W, D = 61, 530, 5
    x0, x1 = smooth_pair(H, W, 8, seed=3)
    x0c, x1c = oracle.cross(x0, 14, 0.05), oracle.cross(x1, 14, 0.05)
    vl, vr = raw_volumes(D, H, W, seed=5)
    for direction, vol in ((-1, vl), (1, vr)):
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol), out, direction, rb=rb, d0=variant, form=8)
        got, want = out.cpu().numpy(), oracle.cbca(x0c, x1c, vol, direction)
        assert same_bits(got, want), diff_report(got, want, "rb=%d variant=%d dir=%d" % (rb, variant, direction))


@pytest.mark.parametrize("H,W,D", [(61, 530, 5), (90, 300, 9), (5, 7, 3), (17, 257, 9), (3, 1030, 5), (140, 130, 3), (1, 9, 2)])
@pytest.mark.parametrize("rb", [2, 4, 8])
@pytest.mark.parametrize("variant", [0, 16, 4, 20, 16 + 96])   # bit 4: one band of rows per XCD, bit 2: own list launch, bits 5 / 6: non-temporal loads / stores
@pytest.mark.parametrize("mk,L1,tau1", [("smooth", 14, 0.05), ("blocky", 14, 0.2)])
def test_lean_wave_geometries(mc, oracle, H, W, D, rb, variant, mk, L1, tau1):
    """a wave per rb output rows x 256 columns, dispatched in address order / per-XCD bands; images smaller than eight bands, ragged
    widths, both directions, a second volume out of the same list"""
    x0, x1 = pair(mk, H, W, D)
    x0c, x1c = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=5)
    v2l, v2r = raw_volumes(D, H, W, seed=6)
    for direction, vol, vol2 in ((-1, vl, v2l), (1, vr, v2r)):
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol), out, direction, rb=rb, d0=variant, nt=H & 1, form=8)
        got, want = out.cpu().numpy(), oracle.cbca(x0c, x1c, vol, direction)
        assert same_bits(got, want), diff_report(got, want, "rb=%d variant=%d dir=%d" % (rb, variant, direction))
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol2), out, direction, rb=rb, d0=variant, form=9)
        got, want = out.cpu().numpy(), oracle.cbca(x0c, x1c, vol2, direction)
        assert same_bits(got, want), diff_report(got, want, "reading the list: rb=%d variant=%d dir=%d" % (rb, variant, direction))


def test_lean_special_values(mc, oracle):
    """zeros, negative zeros, denormals, huge values, infinities and NaNs inside the valid region: the lean kernel's window reads
    every neighbour, but a value outside an output's support is an operand only of outputs that are listed and recomputed; a
    3 x 3 support of nothing but -0.0 sums to +0.0 (adcensus.cu:356)"""
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
    with np.errstate(all="ignore"):
        want = oracle.cbca(x0c, x1c, vl, -1)
    for nt in (0, 1):
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, nt=nt, form=8)
        got = out.cpu().numpy()
        assert same_bits(got, want), diff_report(got, want, "lean + list, special values nt=%d" % nt)


def test_lean_falls_back_to_the_strip_kernel(mc, oracle):
    """(a) a list written for another problem (other direction / other shape on the same cached scratch) is not used;
    (b) a list that cannot hold the pair's entries (the hook's nd = capacity in 16-byte slots), (c) a list written for other rows
    per wave -- the strip kernel runs instead, results stay exact"""
    H, W, D = 33, 140, 4
    x0, x1 = smooth_pair(H, W, 8, seed=2)
    x0c, x1c = oracle.cross(x0, 14, 0.05), oracle.cross(x1, 14, 0.05)
    vl, vr = raw_volumes(D, H, W, seed=5)
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, form=8)
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vr), out, 1, form=9)     # the list on the scratch is direction -1's
    got, want = out.cpu().numpy(), oracle.cbca(x0c, x1c, vr, 1)
    assert same_bits(got, want), diff_report(got, want, "list of the other direction")
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, rb=4, form=8)
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, rb=8, form=9)   # the list on the scratch was written for 4 rows per wave
    got, want = out.cpu().numpy(), oracle.cbca(x0c, x1c, vl, -1)
    assert same_bits(got, want), diff_report(got, want, "list of another wave geometry")
    for Hf, Wf, Df in ((64, 300, 2), (200, 600, 1)):
        z = np.zeros((Hf, Wf), np.float32)
        zc = oracle.cross(z, 14, 1.0)
        v, _ = raw_volumes(Df, Hf, Wf, seed=6)
        want = oracle.cbca(zc, zc, v, -1)
        for form in (8, 9):
            out = torch.full((1, Df, Hf, Wf), -7.0, device="cuda")
            mc.adcensus.cbca_cfg(dev(zc), dev(zc), dev(v), out, -1, nd=1000, form=form)
            got = out.cpu().numpy()
            assert same_bits(got, want), diff_report(got, want, "every output listed, form %d" % form)


def test_fused_predict_on_a_texture_takes_the_lean_path(mc, oracle):
    """mc_predict with the Middlebury parameter set (2 + 16 passes, L1 = 14) on a textured pair: the route word picks the strip
    kernel's route, which the fused path serves with the lean + list kernels -- all five outputs against the oracle"""
    H, W, D = 70, 420, 24
    prm = dict(mc.PRESETS["mb_slow"], cbca_i2=3)
    x0, x1 = smooth_pair(H, W, 10, seed=9)
    vl, vr = raw_volumes(D, H, W, seed=5)
    want = oracle.stereo_predict(prm, x0, x1, D, rawL=vl, rawR=vr)
    xb = dev(np.stack([x0, x1]))[:, None]
    got = mc.stereo_predict_fused(xb, prm, D, raw=(dev(vl), dev(vr)), want_volumes=True, want_disp0=True)
    for k in ("volL", "volR", "dispL0", "dispR0", "disp"):
        g = got[k].cpu().numpy()
        assert same_bits(g, want[k]), diff_report(g, want[k], k)
