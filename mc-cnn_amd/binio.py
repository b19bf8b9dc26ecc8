"""`.bin` outputs of `main.lua -a predict`: raw little-endian float32, no header,
C order -- left.bin/right.bin are (1,D,H,W), disp.bin is (1,1,H,W)
(main.lua:1045,1103; samples/load_bin.py:3-5)."""
import numpy as np


def write_bin(path, array):
    a = np.ascontiguousarray(np.asarray(array), dtype="<f4")
    a.tofile(path)
    return a.shape


def read_bin(path, shape):
    a = np.fromfile(path, dtype="<f4")
    return a.reshape(shape)
