"""CPU: mc_read_png16 / mc_write_png16 / mc_write_pfm (the Lua shim's adcensus.readPNG16 / writePNG16 / writePFM) against what the
REFERENCE'S OWN functions produced (adcensus.cu:1670-1721; tests/golden/make_golden_hostio.py ran them through oracle/_ref -- they are host
code -- with png++ replaced by a container that keeps the pixels): the float -> 16-bit pixel conversion on its cuts, the pixel -> float
conversion on EVERY 16-bit value, the PFM file byte for byte.  The PNG container itself is libpng's on the reference's side: files are
exchanged with libpng (PIL) here."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
Image = pytest.importorskip("PIL.Image")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hostio_ref.npz")


def test_write_png16_stores_the_reference_s_pixels(mc, tmp_path):
    g = np.load(GOLD)
    img, want = g["img"], g["pixels"]
    H, W = img.shape
    p = str(tmp_path / "w.png")
    mc.adcensus.writePNG16(torch.from_numpy(img), H, W, p)
    got = np.asarray(Image.open(p))                      # decoded by libpng
    assert got.dtype == np.uint16 and np.array_equal(got, want)
    assert (want == 0).sum() >= 5 and want.max() == 65535 and (want[2] == 256 * np.arange(W) + 255).all()   # the cuts are in the vector: (k + 255.5 / 256) * 256 is truncated, not rounded


def test_read_png16_returns_the_reference_s_floats_for_every_pixel_value(mc, tmp_path):
    g = np.load(GOLD)
    every, want = g["every"], g["floats"]
    p = str(tmp_path / "every.png")
    Image.fromarray(every).save(p, format="PNG")         # encoded by libpng (its own filter choice per row)
    got = torch.full(every.shape, -7.0, dtype=torch.float32)
    assert mc.adcensus.readPNG16(got, p) == every.shape
    assert got.numpy().tobytes() == want.tobytes()
    assert want[0, 0] == 0 and want[0, 1] == np.float32(1 / 256) and want[255, 255] == np.float32(65535 / 256)


def test_write_pfm_is_the_reference_s_file(mc, tmp_path):
    g = np.load(GOLD)
    p = str(tmp_path / "x.pfm")
    mc.adcensus.writePFM(torch.from_numpy(g["img"]), p)
    assert open(p, "rb").read() == g["pfm"].tobytes()


def test_against_the_reference_library_when_built(mc, tmp_path):
    try:
        from oracle.ref_lib import RefLib
        ref = RefLib()
        assert "adcensus.writePNG16" in ref.functions()
    except Exception as e:   # (a box without oracle/_ref: the golden vectors above are the pin)
        if os.environ.get("MC_REQUIRE_REF") == "1" and os.path.exists("/root/reference/adcensus.cu"):
            raise
        pytest.skip("oracle/_ref not built: %s" % e)
    rng = np.random.default_rng(23)
    img = (rng.random((19, 33)) * 256).astype(np.float32)
    img[0, :4] = [0, 1e-5, 255.999, 65535 / 256]
    rp, cp = str(tmp_path / "r.raw16"), str(tmp_path / "c.png")
    try:
        ref.call("writePNG16", torch.from_numpy(img), 19, 33, rp)
    except Exception as e:   # an oracle/_ref built with the inert png++ stand-in of earlier rounds
        pytest.skip("oracle/_ref predates the pixel-keeping png++ stand-in: %s" % e)
    mc.adcensus.writePNG16(torch.from_numpy(img), 19, 33, cp)
    body = open(rp, "rb").read().split(b"\n", 1)[1]
    assert np.array_equal(np.frombuffer(body, "<u2").reshape(19, 33), np.asarray(Image.open(cp)))
    a, b = str(tmp_path / "a.pfm"), str(tmp_path / "b.pfm")
    ref.call("writePFM", torch.from_numpy(img), a)
    mc.adcensus.writePFM(torch.from_numpy(img), b)
    assert open(a, "rb").read() == open(b, "rb").read()
