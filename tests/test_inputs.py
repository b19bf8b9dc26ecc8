"""The synthetic image pairs of bench.py / the tests: their cross-arm statistics are what decides the cost of cbca."""
import numpy as np

from util import natural_pair, smooth_pair


def arm_lengths(oracle, x, L1, tau1):
    H, W = x.shape
    c = np.asarray(oracle.cross(x, L1, tau1)).reshape(4, H, W)   # exclusive end coordinates (adcensus.cu:280-322)
    xs, ys = np.arange(W)[None, :], np.arange(H)[:, None]
    return np.stack([xs - c[0] - 1, c[1] - xs - 1, ys - c[2] - 1, c[3] - ys - 1]).astype(int)


def non_minimal_share(a):
    own = (a == 1).all(0)
    lr1 = (a[0] == 1) & (a[1] == 1)
    return 1.0 - (own & np.roll(lr1, 1, 0) & np.roll(lr1, -1, 0))[2:-2, 2:-2].mean()


def test_natural_pair_has_real_scene_arm_statistics(oracle):
    """calibration targets (the reference's sample pair, measured in the build container; tests/util.natural_pair):
    cross(5, 0.13): ~90 % of the supports larger than 3x3, 60 % of a single image's arms at the limit 4;
    cross(14, 0.02): about half of the supports larger than 3x3, a few per cent of the arms at the limit 13"""
    x0, x1 = natural_pair(200, 600, 64, seed=3)
    both = lambda L1, tau1: np.minimum(arm_lengths(oracle, x0, L1, tau1), arm_lengths(oracle, x1, L1, tau1))   # as cbca at d = 0
    a = both(5, 0.13)
    assert 0.8 < non_minimal_share(a) < 0.97
    assert 0.3 < (a == 4).mean() < 0.75
    a = both(14, 0.02)
    assert 0.3 < non_minimal_share(a) < 0.7
    assert 0.005 < (a == 13).mean() < 0.15
    assert abs(x0.mean()) < 1e-3 and abs(x0.std() - 1) < 1e-2 and x0.dtype == np.float32


def test_smooth_pair_is_the_textured_extreme(oracle):
    x0, x1 = smooth_pair(120, 400, 32, seed=3)
    a = np.minimum(arm_lengths(oracle, x0, 14, 0.02), arm_lengths(oracle, x1, 14, 0.02))
    assert non_minimal_share(a) < 0.1 and a.max() <= 13
