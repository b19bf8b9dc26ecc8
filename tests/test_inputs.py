"""The synthetic image pairs of bench.py / the tests: their cross-arm statistics are what decides the cost of cbca."""
import numpy as np

from util import mixed_pair, natural_pair, sample_pair, smooth_pair


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


def test_mixed_pair_is_a_texture_with_flat_patches(oracle):
    """tests/util.mixed_pair (bench.py's `north_star.mixed_pair`): the Gaussian texture's minimal supports outside the patches, arms at the
    L1 - 1 limit inside them (exactly constant in BOTH images), the patches over the asked share of the image; seeded"""
    H, W, D = 300, 900, 64
    x0, x1 = mixed_pair(H, W, D, seed=7, flat_frac=0.15)
    y0, y1 = mixed_pair(H, W, D, seed=7, flat_frac=0.15)
    assert np.array_equal(x0, y0) and np.array_equal(x1, y1)
    assert abs(x0.mean()) < 1e-3 and abs(x0.std() - 1) < 1e-2 and x0.dtype == np.float32
    flat = x0 == x0.max()                                    # (the patches are the clipping level)
    assert 0.14 < flat.mean() < 0.25
    a0, a1 = arm_lengths(oracle, x0, 14, 0.02), arm_lengths(oracle, x1, 14, 0.02)
    from scipy.ndimage import binary_erosion
    deep = binary_erosion(flat, structure=np.ones((27, 27), bool))   # 13 flat pixels in every direction: all four arms of the left image at the limit
    assert deep.any() and (a0[:, deep] == 13).all()
    assert (a1 == 13).mean() > 0.02                          # ... and the right image has them too (shifted)
    tex = ~flat
    tex[:2] = tex[-2:] = False; tex[:, :2] = tex[:, -2:] = False
    assert (a0[:, tex] == 1).mean() > 0.8                    # outside: the texture's unit arms
    s4 = mixed_pair(H, W, D, seed=7, flat_frac=0.04)[0]
    assert 0.03 < (s4 == s4.max()).mean() < 0.12


def arm_stats(oracle, x0, x1, L1, tau1, ds=(0, 30, 60, 120)):
    """share of supports larger than 3x3 / share of per-arm minima at the L1 - 1 limit / mean arm, arms combined over both
    images as cbca combines them (left pixel x with right pixel x - d)"""
    a0, a1 = arm_lengths(oracle, x0, L1, tau1), arm_lengths(oracle, x1, L1, tau1)
    W = x0.shape[1]
    rows = []
    for d in ds:
        a = np.minimum(a0[:, :, d:], a1[:, :, :W - d] if d else a1)
        rows.append((non_minimal_share(a), (a == L1 - 1).mean(), a.mean()))
    return np.array(rows)


def test_natural_pair_is_calibrated_against_the_reference_sample_pair(oracle):
    """tests/util.natural_pair (the synthetic pair of the 1000x1500 realistic record and of most GPU tests) against the
    reference's real pair under the three parameter sets its docstring quotes: the statistics that decide the cost of
    cbca -- share of non-minimal supports, share of arms at the limit, mean arm -- agree within the stated bands"""
    real = sample_pair()
    syn = natural_pair(370, 1226, 228)
    for (L1, tau1), tol in (((5, 0.13), (0.08, 0.15, 0.5)), ((14, 0.02), (0.12, 0.06, 0.8)), ((5, 0.03), (0.15, 0.12, 0.5))):
        r, s_ = arm_stats(oracle, *real, L1, tau1), arm_stats(oracle, *syn, L1, tau1)
        for k in range(3):
            assert abs(r[:, k].mean() - s_[:, k].mean()) < tol[k], (L1, tau1, k, r[:, k], s_[:, k])


def test_sample_pair_fixture():
    x0, x1 = sample_pair()
    assert x0.shape == x1.shape == (370, 1226) and x0.dtype == np.float32
    assert abs(x0.mean()) < 1e-3 and abs(x0.std(ddof=1) - 1) < 1e-3
    t0, t1 = sample_pair(1000, 1500)
    assert t0.shape == (1000, 1500) and abs(t1.mean()) < 1e-3


def test_tile_step_model_runs():
    """scripts/model/tile_steps.py (the CPU model the tile kernel's leads were ranked with) at a toy size"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "model", "tile_steps.py"), "natural", "5", "--reg3", "--size=128x384x32"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "mean critical path" in r.stdout and "work share" in r.stdout
