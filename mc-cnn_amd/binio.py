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


_TYPES = {"float32": "<f4", "int32": "<i4", "int64": "<i8"}


def fromfile(fname):
    """main.lua:353-380: a raw array `fname` with sidecars `fname.dim` (one extent per line) and `fname.type`
    (float32 | int32 | int64) -- the dataset format written by preprocess_kitti.lua:118-134 / preprocess_mb.py:99-106."""
    dim = [int(float(line)) for line in open(fname + ".dim").read().split()]
    if dim == [0]:
        return np.zeros((0,), np.float32)
    t = open(fname + ".type").read().strip()
    if t not in _TYPES:
        raise ValueError("%s: unknown element type %r" % (fname, t))
    return np.fromfile(fname, dtype=_TYPES[t]).reshape(dim)


def tofile(fname, array):
    """The writer side of `fromfile` (preprocess_mb.py:99-106)."""
    a = np.ascontiguousarray(array)
    t = {np.dtype("float32"): "float32", np.dtype("int32"): "int32", np.dtype("int64"): "int64"}[a.dtype]
    a.astype(_TYPES[t]).tofile(fname)
    open(fname + ".type", "w").write(t)
    open(fname + ".dim", "w").write("\n".join(str(d) for d in a.shape) + "\n")


# ---- submission formats, adcensus.cu:1670-1721 (host side of libadcensus) ---------------------------------------------------
def read_png16(fname):
    """adcensus.readPNG16: 16-bit grey PNG -> float disparities, val/256 with 0 kept as 0 (KITTI ground truth)."""
    from PIL import Image
    a = np.asarray(Image.open(fname))
    if a.dtype != np.uint16:
        a = a.astype(np.uint16)
    return np.where(a == 0, np.float32(0.0), a.astype(np.float32) / np.float32(256.0)).astype(np.float32)


def write_png16(img, fname):
    """adcensus.writePNG16: (uint16)(val < 1e-5 ? 0 : val * 256) per pixel (float multiply, truncation)."""
    from PIL import Image
    v = np.asarray(img, np.float32)
    q = np.where(v < np.float32(1e-5), np.float32(0.0), v * np.float32(256.0))
    Image.fromarray(q.astype(np.uint16)).save(fname, format="PNG")


def write_pfm(img, fname):
    """adcensus.writePFM: 'Pf', 'W H', scale -0.003922 (little-endian), rows as stored (no flip), adcensus.cu:1705-1719."""
    a = np.ascontiguousarray(np.asarray(img), dtype="<f4")
    h, w = a.shape
    with open(fname, "wb") as f:
        f.write(("Pf\n%d %d\n-0.003922\n" % (w, h)).encode("ascii"))
        f.write(a.tobytes())
