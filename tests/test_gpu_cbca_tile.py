"""-m gpu: the LDS-tile kernel of cross-based aggregation (cbca forms 2 / 3 of the hook: short-arm instance, arms <= 4, and
long-arm instance, arms <= 13 -- what mc_predict runs for L1 <= 5 / L1 <= 14) against the oracle, bit for bit: every kind of
arm statistics, tiles with ragged edges (H, W not multiples of the tile), images smaller than one tile, both directions, both
cache policies, plane sub-ranges, special values inside the valid region."""
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


SHAPES = [(90, 300, 9), (41, 519, 6), (27, 253, 5), (83, 64, 12), (37, 449, 4), (140, 130, 3), (5, 7, 3), (16, 128, 8), (17, 129, 9)]


@pytest.mark.parametrize("H,W,D", SHAPES)
@pytest.mark.parametrize("mk,L1,tau1", [("smooth", 5, 0.13), ("random", 5, 0.5), ("blocky", 5, 0.2), ("natural", 5, 0.13),
                                        ("natural", 3, 0.03), ("blocky", 2, 0.3), ("smooth", 0, 0.0), ("flat", 5, 1.0)])
def test_tile_kernel_short_arms(mc, oracle, H, W, D, mk, L1, tau1):
    x0, x1 = pair(mk, H, W, D)
    x0c, x1c = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=13)
    for direction, vol in ((-1, vl), (1, vr)):
        want = oracle.cbca(x0c, x1c, vol, direction)
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol), out, direction, nt=(H + W) & 1, form=2)
        got = out.cpu().numpy()
        assert same_bits(got, want), diff_report(got, want, "tile kernel (arms <= 4) dir=%d" % direction)


@pytest.mark.parametrize("H,W,D", SHAPES)
@pytest.mark.parametrize("mk,L1,tau1", [("smooth", 14, 0.02), ("natural", 14, 0.02), ("blocky", 14, 0.2), ("natural", 9, 0.05),
                                        ("blocky", 6, 0.3), ("natural", 5, 0.13), ("random", 14, 2.5), ("flat", 14, 1.0),
                                        ("flat", 11, 1.0)])
def test_tile_kernel_long_arms(mc, oracle, H, W, D, mk, L1, tau1):
    x0, x1 = pair(mk, H, W, D)
    x0c, x1c = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=13)
    for direction, vol in ((-1, vl), (1, vr)):
        want = oracle.cbca(x0c, x1c, vol, direction)
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol), out, direction, nt=(H + W) & 1, form=3)
        got = out.cpu().numpy()
        assert same_bits(got, want), diff_report(got, want, "tile kernel (arms <= 13) dir=%d" % direction)


@pytest.mark.parametrize("form,L1", [(2, 5), (3, 14)])
def test_tile_kernel_plane_range(mc, oracle, form, L1):
    H, W, D = 30, 200, 21
    x0, x1 = blocky_pair(H, W, seed=4)
    x0c, x1c = oracle.cross(x0, L1, 0.2), oracle.cross(x1, L1, 0.2)
    vl, _ = raw_volumes(D, H, W, seed=2)
    want = oracle.cbca(x0c, x1c, vl, -1)
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, d0=3, nd=13, form=form)
    got = out.cpu().numpy()[0]
    assert same_bits(got[3:16], want[3:16]), diff_report(got[3:16], want[3:16], "planes 3..15")
    assert (got[:3] == -7.0).all() and (got[16:] == -7.0).all(), "planes outside [d0, d0+nd) were written"


@pytest.mark.parametrize("form,L1", [(2, 5), (3, 14)])
def test_tile_kernel_special_values(mc, oracle, form, L1):
    """zeros, negative zeros, denormals, huge values, infinities and NaNs inside the valid region: a value that is not in a
    support is never an operand of its chain -- neither a neighbour's tap nor what an accumulator collected before its
    output's first row -- and a support of nothing but -0.0 sums to +0.0 (adcensus.cu:356)"""
    H, W, D = 40, 260, 6
    x0, x1 = blocky_pair(H, W, seed=8)
    x0c, x1c = oracle.cross(x0, L1, 0.2), oracle.cross(x1, L1, 0.2)
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
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, nt=nt, form=form)
        got = out.cpu().numpy()
        assert same_bits(got, want), diff_report(got, want, "tile kernel form %d, special values nt=%d" % (form, nt))


def test_plan_of_another_problem_is_not_read(mc, oracle):
    """forms 6 / 7 on a scratch whose plan was written for the other direction (or never): the pass stands down -- nothing is
    written -- instead of walking a foreign plan (the plan's head says who wrote it)"""
    H, W, D = 40, 150, 5
    x0, x1 = blocky_pair(H, W, seed=2)
    vl, vr = raw_volumes(D, H, W, seed=3)
    for L1, forms in ((5, (4, 6)), (14, (5, 7))):
        x0c, x1c = oracle.cross(x0, L1, 0.2), oracle.cross(x1, L1, 0.2)
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, form=forms[0])
        assert same_bits(out.cpu().numpy(), oracle.cbca(x0c, x1c, vl, -1))
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vr), out, 1, form=forms[1])      # the plan on the scratch is direction -1's
        assert (out.cpu().numpy() == -7.0).all()
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, form=forms[1])     # ... and is still direction -1's
        assert same_bits(out.cpu().numpy(), oracle.cbca(x0c, x1c, vl, -1))
    x0c, x1c = oracle.cross(x0, 5, 0.2), oracle.cross(x1, 5, 0.2)                  # arm class: a plan written by the long-arm instance
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, form=5)
    out = torch.full((1, D, H, W), -7.0, device="cuda")
    mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, form=6)
    assert (out.cpu().numpy() == -7.0).all()


def test_tile_kernel_stands_down_when_an_arm_is_too_long(mc, oracle):
    """the hook's forms 2 / 3 write nothing if cbca_pack saw an arm beyond the instance's class"""
    H, W, D = 40, 100, 3
    x0 = np.zeros((H, W), np.float32)
    x0c = oracle.cross(x0, 20, 1.0)
    vl, _ = raw_volumes(D, H, W, seed=3)
    for form in (2, 3):
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x0c), dev(vl), out, -1, form=form)
        assert (out.cpu().numpy() == -7.0).all()


PLAN_SHAPES = [(90, 300, 9), (41, 519, 6), (83, 64, 12), (140, 130, 3), (5, 7, 3), (17, 129, 9), (200, 1100, 4)]


@pytest.mark.parametrize("H,W,D", PLAN_SHAPES)
@pytest.mark.parametrize("mk,L1,tau1,forms", [("natural", 5, 0.13, (4, 6)), ("blocky", 5, 0.2, (4, 6)), ("smooth", 5, 0.13, (4, 6)),
                                              ("natural", 14, 0.02, (5, 7)), ("blocky", 14, 0.2, (5, 7)), ("smooth", 14, 0.02, (5, 7)),
                                              ("flat", 14, 1.0, (5, 7)), ("natural", 9, 0.05, (5, 7))])
def test_tile_kernel_with_plan(mc, oracle, H, W, D, mk, L1, tau1, forms):
    """what mc_predict does from the second aggregation pass of a direction on: the pass that sorts also writes every step's
    item order (forms 4 / 5), later passes over OTHER volumes of the same pair read it (forms 6 / 7) -- each bit-identical to the
    oracle; both directions (each with its own plan), a plane sub-range of a full plan"""
    x0, x1 = pair(mk, H, W, D)
    x0c, x1c = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=13)
    v2l, v2r = raw_volumes(D, H, W, seed=14)
    for direction, vol, vol2 in ((-1, vl, v2l), (1, vr, v2r)):
        out = torch.full((1, D, H, W), -7.0, device="cuda")
        mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol), out, direction, form=forms[0])
        got, want = out.cpu().numpy(), oracle.cbca(x0c, x1c, vol, direction)
        assert same_bits(got, want), diff_report(got, want, "tile kernel writing the plan, dir=%d" % direction)
        for v in (vol2, vol):
            out = torch.full((1, D, H, W), -7.0, device="cuda")
            mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(v), out, direction, nt=(H + W) & 1, form=forms[1])
            got, want = out.cpu().numpy(), oracle.cbca(x0c, x1c, v, direction)
            assert same_bits(got, want), diff_report(got, want, "tile kernel reading the plan, dir=%d" % direction)
        if D >= 6:
            out = torch.full((1, D, H, W), -7.0, device="cuda")
            mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vol2), out, direction, d0=2, nd=3, form=forms[1])
            got = out.cpu().numpy()[0]
            want = oracle.cbca(x0c, x1c, vol2, direction)
            assert same_bits(got[2:5], want[2:5]) and (got[:2] == -7.0).all() and (got[5:] == -7.0).all()
