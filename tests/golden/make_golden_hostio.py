#!/usr/bin/env python
"""tests/golden/make_golden_hostio.py -- tests/golden/hostio_ref.npz from the REFERENCE ITSELF: adcensus.readPNG16 / writePNG16 / writePFM
(/root/reference/adcensus.cu:1670-1721) are host code, so the reference library of oracle/build_ref.py runs them in the build container
without a GPU.  png++ / libpng (third-party, absent from /root/reference) are replaced by oracle/ref_stubs/png++/image.hpp, which keeps
the 16-bit pixels in a trivial container ("MCREF16 w h\\n" + little-endian uint16) -- what the reference's source defines, the
float <-> pixel arithmetic, runs as written.

    python oracle/build_ref.py && python tests/golden/make_golden_hostio.py

Holds: a float image with the cuts of writePNG16's conversion (0, 1e-5, exact counts, the top of the range, values between counts) and
the reference's pixels for it; EVERY 16-bit pixel value and the reference's floats for it (readPNG16); the reference's PFM bytes."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ref_lib import RefLib  # noqa: E402


def raw16_write(path, px):
    with open(path, "wb") as f:
        f.write(b"MCREF16 %d %d\n" % (px.shape[1], px.shape[0]))
        f.write(np.ascontiguousarray(px, "<u2").tobytes())


def raw16_read(path):
    b = open(path, "rb").read()
    head, body = b.split(b"\n", 1)
    _, w, h = head.split()
    return np.frombuffer(body, "<u2").reshape(int(h), int(w)).copy()


def main():
    ref = RefLib()
    rng = np.random.default_rng(17)
    H, W = 41, 67
    img = (rng.random((H, W)) * 255.9).astype(np.float32)
    cuts = np.array([0.0, -0.0, 1e-6, 9.9e-6, 1e-5, 1.1e-5, 0.0039, 0.00390625, 0.00390626, 0.5, 1.0, 127.99, 128.0, 255.99609375, 255.996, 255.9999, 3.99609375], np.float32)
    img.ravel()[:cuts.size] = cuts
    img[2, :] = (np.arange(W, dtype=np.float32) + np.float32(0.998046875))    # just below a count boundary: truncation, not rounding
    img[3, :] = np.float32(1) / np.float32(256) * np.arange(W, dtype=np.float32)
    every = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "w.raw16")
        ref.call("writePNG16", torch.from_numpy(img), H, W, p)
        pixels = raw16_read(p)
        q = os.path.join(d, "r.raw16")
        raw16_write(q, every)
        floats = torch.full((256, 256), -7.0, dtype=torch.float32)
        ref.call("readPNG16", floats, q)
        f = os.path.join(d, "x.pfm")
        ref.call("writePFM", torch.from_numpy(img), f)
        pfm = np.frombuffer(open(f, "rb").read(), np.uint8).copy()
    out = os.path.join(ROOT, "tests", "golden", "hostio_ref.npz")
    np.savez_compressed(out, img=img, pixels=pixels, every=every, floats=floats.numpy(), pfm=pfm)
    print(out, pixels.shape, int(pixels.max()), float(floats.numpy().max()), pfm.size)


if __name__ == "__main__":
    main()
