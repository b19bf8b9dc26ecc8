"""CPU: the host side of `main.py` (main.lua's flag tables, image normalisation, seeded nets) -- no GPU needed."""
import numpy as np

from mc_cnn_amd import main as mcmain
from mc_cnn_amd import params


def test_default_tables_cover_main_lua():
    """12 (dataset, arch) tables (main.lua:68-295); spot values from the ones BASELINE.json's configs use."""
    assert len(params.TABLES) == 12
    t = params.TABLES[("kitti", "fast")]
    assert (t["pi1"], t["pi2"], t["sgm_q1"], t["sgm_q2"], t["alpha1"], t["tau_so"], t["blur_sigma"], t["blur_t"]) == \
        (4.0, 55.72, 3.0, 2.5, 1.5, 0.02, 7.74, 5.0)
    t = params.TABLES[("mb", "slow")]
    assert (t["L1"], t["tau1"], t["cbca_i1"], t["cbca_i2"], t["lr_check"], t["border_n"]) == (14, 0.02, 2, 16, 0, 5)
    t = params.TABLES[("kitti2015", "slow")]
    assert (t["cbca_i2"], t["tau1"], t["alpha1"]) == (4, 0.03, 1.75)
    assert params.NET_SHAPES[("mb", "fast")] == (5, 64)


def test_flags_override_defaults_and_stage_names():
    _, _, opt, prm = mcmain.parse(["kitti", "fast", "-a", "predict", "-left", "l.png", "-right", "r.png", "-disp_max", "70",
                                   "-pi1", "3.5", "-sgm_i", "2"])
    assert opt.disp_max == 70 and prm["pi1"] == 3.5 and prm["sgm_i"] == 2 and prm["pi2"] == 55.72
    p = params.make_params(dict(prm, sm_terminate="sgm", sm_skip="median"))
    assert (p.sm_terminate, p.sm_skip) == (3, 5)


def test_normalize_is_torch_unbiased_std():
    rng = np.random.default_rng(0)
    x = (rng.random((1, 7, 9)) * 255).astype(np.float32)
    y = mcmain.normalize(x)
    assert abs(float(y.mean())) < 1e-5
    assert abs(float(y.astype(np.float64).std(ddof=1)) - 1.0) < 1e-5


def test_rgb2y_and_random_net_are_deterministic():
    img = np.stack([np.full((2, 2), 10.0), np.full((2, 2), 20.0), np.full((2, 2), 30.0)]).astype(np.float32)
    assert np.allclose(mcmain.rgb2y(img), 0.299 * 10 + 0.587 * 20 + 0.114 * 30)
    a = mcmain.load_net("random:5", "kitti", "fast")
    b = mcmain.load_net("random:5", "kitti", "fast")
    assert len(a) == 4 and a[0][0].shape == (64, 1, 3, 3) and a[1][0].shape == (64, 64, 3, 3)
    assert all(np.array_equal(x[0], y[0]) for x, y in zip(a, b))
