"""CPU: adcensus.grey2jet (adcensus.cu:2000-2053; host code in the reference too -- the debug images of main.lua:503,1242,1260) behind
the C ABI (mc_grey2jet, bound in Lua and Python under the reference's name): against a golden vector produced by the REFERENCE'S OWN
function (tests/golden/make_golden_grey2jet.py: every boundary of the five pieces from both sides, the ends of the domain, the values
main.lua feeds it), against the oracle's restatement, against the reference library itself where it is built, and on the paths where the
reference asserts."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grey2jet.npz")


def test_grey2jet_equals_the_reference_bit_for_bit(mc, oracle):
    g = np.load(GOLD)
    grey, want = g["grey"], g["col"]
    H, W = grey.shape
    col = torch.full((1, 3, H, W), -7.0, dtype=torch.float64)          # main.lua:1241: torch.Tensor(1, 3, H, W)
    mc.adcensus.grey2jet(torch.from_numpy(grey), col)
    assert col.numpy()[0].tobytes() == want.tobytes()
    assert oracle.grey2jet(grey).tobytes() == want.tobytes()             # the oracle's restatement is pinned by the same vector
    # the pieces are all there, and a colour plane is a plane of the output (red first)
    val = 4 * grey
    for lo, hi in ((-0.1, 0.5), (0.5, 1.5), (1.5, 2.5), (2.5, 3.5), (3.5, 4.1)):
        assert ((val >= lo) & (val < hi)).any()
    assert np.array_equal(want[0][val < 1.5], np.zeros((val < 1.5).sum())) and np.array_equal(want[2][(val >= 2.5)], np.zeros((val >= 2.5).sum()))


def test_grey2jet_against_the_reference_library_when_built(mc):
    try:
        from oracle.ref_lib import RefLib, RefUnavailable
        ref = RefLib()
    except Exception as e:   # (the GPU box and CI without /root/reference: the golden vector above is the pin)
        if os.environ.get("MC_REQUIRE_REF") == "1" and os.path.exists("/root/reference/adcensus.cu"):
            raise
        pytest.skip("oracle/_ref not built: %s" % e)
    rng = np.random.default_rng(3)
    for H, W in ((1, 1), (7, 13), (40, 64)):
        grey = torch.from_numpy(rng.uniform(-0.025, 1.025, (H, W)))
        a = torch.full((3, H, W), -1.0, dtype=torch.float64)
        b = torch.full((3, H, W), -2.0, dtype=torch.float64)
        mc.adcensus.grey2jet(grey, a)
        ref.call("grey2jet", grey, b)
        assert a.numpy().tobytes() == b.numpy().tobytes()


def test_grey2jet_reports_what_the_reference_asserts_on(mc, oracle):
    for bad in (1.03, -0.03, float("nan"), float("inf")):
        grey = torch.full((3, 4), 0.5, dtype=torch.float64)
        grey[1, 2] = bad
        col = torch.full((3, 3, 4), -7.0, dtype=torch.float64)
        with pytest.raises(mc._lib.McError, match=r"grey2jet.*\(1, 2\)"):
            mc.adcensus.grey2jet(grey, col)
        assert col[0, 0, 0].item() == 0.5 and col[1, 0, 0].item() == 1.0 and col[0, 2, 3].item() == -7.0   # pixels before the offending one are written, as in the reference
        with pytest.raises(ValueError):
            oracle.grey2jet(grey.numpy())
    with pytest.raises(ValueError, match="Size mismatch"):                   # adcensus.cu:2007-2009
        mc.adcensus.grey2jet(torch.zeros((3, 4), dtype=torch.float64), torch.zeros((2, 3, 4), dtype=torch.float64))
    with pytest.raises(AssertionError):
        mc.adcensus.grey2jet(torch.zeros((3, 4), dtype=torch.float32), torch.zeros((3, 3, 4), dtype=torch.float64))
