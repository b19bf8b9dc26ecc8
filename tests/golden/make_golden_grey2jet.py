#!/usr/bin/env python
"""tests/golden/make_golden_grey2jet.py -- tests/golden/grey2jet.npz from the REFERENCE ITSELF: adcensus.grey2jet of
/root/reference/adcensus.cu:2000-2053 is host code ("CPU implementation", torch.DoubleTensor arguments), so the reference library
built by oracle/build_ref.py runs it in the build container without a GPU.

    python oracle/build_ref.py && python tests/golden/make_golden_grey2jet.py

The file holds the input and the reference's output: tests/test_grey2jet.py pins mc_grey2jet and the oracle's restatement to it, bit for
bit, with neither a GPU nor /root/reference."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ref_lib import RefLib  # noqa: E402


def main():
    ref = RefLib()
    rng = np.random.default_rng(11)
    H, W = 23, 61
    g = rng.uniform(-0.025, 1.025, (H, W))            # val = 4 g in [-0.1, 4.1]: the whole domain of the map
    # every boundary of the five pieces, from both sides, and the ends of the domain (val = 4 g exactly representable where g is)
    edges = np.array([-0.1, 0.5, 1.5, 2.5, 3.5, 4.1]) / 4
    pts = np.concatenate([edges, np.nextafter(edges, -np.inf)[1:], np.nextafter(edges, np.inf)[:-1], [0.0, -0.0, 0.25, 0.5, 0.75, 1.0, 1.0 / 3, 228 / 228.0, 71 / 70.0]])
    pts = pts[(4 * pts >= -0.1) & (4 * pts <= 4.1)]
    g.ravel()[:pts.size] = pts
    # what main.lua feeds it: (disparity + 1) / disp_max of a float map (main.lua:503,1242)
    d = rng.integers(0, 70, (H,)).astype(np.float32)
    g[:, -1] = (d.astype(np.float64) + 1) / 70
    grey = torch.from_numpy(np.ascontiguousarray(g))
    col = torch.full((1, 3, H, W), -7.0, dtype=torch.float64)
    ref.call("grey2jet", grey, col)
    out = os.path.join(ROOT, "tests", "golden", "grey2jet.npz")
    np.savez_compressed(out, grey=grey.numpy(), col=col.numpy()[0])
    print(out, grey.shape, float(col.min()), float(col.max()))


if __name__ == "__main__":
    main()
