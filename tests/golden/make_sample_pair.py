"""Generator of tests/golden/kitti_sample_pair.npz: the reference's one real input pair, samples/input/kittiL.png /
kittiR.png (370 x 1226, 8-bit grey), as two uint8 arrays.  Run in the build container (the reference tree does not travel
to the GPU box):  python tests/golden/make_sample_pair.py
SURVEY.md section 8(d) specifies the KITTI-shaped configurations on this pair; tests/util.sample_pair() loads and
normalises it the way main.lua:1085-1096 does."""
import os
import sys

import numpy as np
from PIL import Image

REF = "/root/reference/samples/input"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {}
    for key, name in (("left", "kittiL.png"), ("right", "kittiR.png")):
        im = Image.open(os.path.join(REF, name))
        a = np.asarray(im)
        assert a.dtype == np.uint8 and a.ndim == 2, (name, a.dtype, a.shape)   # single-channel 8-bit: main.lua's rgb2y branch is not taken
        out[key] = a
    assert out["left"].shape == out["right"].shape == (370, 1226)
    np.savez_compressed(os.path.join(HERE, "kitti_sample_pair.npz"), **out)
    print({k: (v.shape, int(v.min()), int(v.max()), float(v.mean())) for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
