"""Seeded synthetic inputs shared by the parity tests (shapes follow SURVEY.md section 8d)."""
import numpy as np


def normalize_image(img):
    """x:add(-x:mean()):div(x:std()) with torch's unbiased std, main.lua:1095-1096."""
    img = img.astype(np.float64)
    return ((img - img.mean()) / img.std(ddof=1)).astype(np.float32)


def smooth_pair(H, W, D, seed=1234, noise=0.05):
    """Left = blurred N(0,1) field; right = left shifted by a smooth disparity field + noise."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    left = gaussian_filter(rng.standard_normal((H, W + D)), 3.0)
    disp = gaussian_filter(rng.random((H, W + D)), 12.0)
    disp = (disp - disp.min()) / (disp.max() - disp.min() + 1e-12) * 0.8 * (D - 1)
    xs = np.arange(W + D)[None, :] + disp
    x0 = np.floor(xs).astype(int).clip(0, W + D - 1)
    x1 = (x0 + 1).clip(0, W + D - 1)
    f = xs - np.floor(xs)
    rows = np.arange(H)[:, None]
    right = left[rows, x0] * (1 - f) + left[rows, x1] * f
    right = right + noise * rng.standard_normal(right.shape)
    return normalize_image(left[:, :W]), normalize_image(right[:, :W])


def natural_pair(H, W, D, seed=1234, sigma=40.0, tex_frac=0.25, tex_amp=0.35, sensor_noise=0.3, levels_per_std=60.0, clip=1.3):
    """A pair with the cross-arm statistics of a real 8-bit road scene: large smooth regions, a quarter of the area
    textured, sensor noise of a fraction of a grey level, rounding to grey levels (neighbouring pixels are often EQUAL),
    clipped highlights; right = left shifted by a smooth disparity field.
    Calibrated in the build container against the reference's sample pair (samples/input/kittiL.png / kittiR.png, which
    do not travel): share of outputs whose support is larger than the minimal 3x3 / share of per-arm minima at the L1-1
    limit / mean arm, arms combined over both images at d = 0 .. 120:
                                    real pair                     natural_pair                 smooth_pair
        cross(L1=5,  tau1=0.13)     0.90 / 0.37-0.48 / 2.3-2.6    0.87-0.89 / 0.40-0.47 / 2.5-2.7    0.20 / 0.00 / 1.03
        cross(L1=14, tau1=0.02)     0.44-0.50 / 0.02-0.07 / 1.5-2.2    0.48-0.52 / 0.03-0.05 / 1.8-2.2    0.013 / 0 / 1.00
        cross(L1=5,  tau1=0.03)     0.57-0.60 / 0.12-0.21 / 1.5-1.8    0.48-0.52 / 0.09-0.13 / 1.4-1.5
    (the Gaussian textures of smooth_pair / random_pair are the regime in which nearly every support is the minimal 3x3)"""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    Wp = W + D
    base = gaussian_filter(rng.standard_normal((H, Wp)), sigma)
    base /= base.std()
    sel = gaussian_filter(rng.standard_normal((H, Wp)), 25.0)
    tex = (sel > np.quantile(sel, 1 - tex_frac)).astype(np.float64)
    clean = base + tex * tex_amp * rng.standard_normal((H, Wp))
    disp = gaussian_filter(rng.random((H, Wp)), 12.0)
    disp = (disp - disp.min()) / (disp.max() - disp.min() + 1e-12) * 0.8 * (D - 1)
    xs = np.arange(Wp)[None, :] + disp
    x0 = np.floor(xs).astype(int).clip(0, Wp - 1)
    x1 = (x0 + 1).clip(0, Wp - 1)
    f = xs - np.floor(xs)
    rows = np.arange(H)[:, None]
    shifted = clean[rows, x0] * (1 - f) + clean[rows, x1] * f

    def sensor(im):  # noise of a fraction of a grey level, rounding to grey levels, highlights clipped
        return np.minimum(np.round(im * levels_per_std + sensor_noise * rng.standard_normal(im.shape)), np.round(clip * levels_per_std))
    return normalize_image(sensor(clean)[:, :W]), normalize_image(sensor(shifted)[:, :W])


def mixed_pair(H, W, D, seed=1234, flat_frac=0.15, patch=None, noise=0.05):
    """The regime BETWEEN the two: smooth_pair's Gaussian texture (nearly every support the minimal 3 x 3) with flat patches --
    clipped highlights: exactly constant in both images, no sensor noise -- covering `flat_frac` of the image (10-20 %).
    Inside a patch every arm runs to the L1 limit, so the pair has (2 L1 - 1)^2-tap supports next to 3 x 3 ones; the right image is the
    left one shifted by smooth_pair's disparity field, the patches with it.  patch = side of a patch in pixels (default: H / 12,
    at least 16); patches are dropped at seeded positions until their union covers flat_frac."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    Wp = W + D
    left = gaussian_filter(rng.standard_normal((H, Wp)), 3.0)
    left /= left.std()
    side = int(patch) if patch else max(16, H // 12)
    flat = np.zeros((H, Wp), bool)
    for _ in range(10000):
        if flat[:, :W].mean() >= flat_frac:
            break
        y, x = int(rng.integers(0, max(1, H - side // 2))), int(rng.integers(0, max(1, Wp - side // 2)))
        h, w = int(rng.integers(side // 2, side + 1)), int(rng.integers(side, 2 * side + 1))
        flat[y:y + h, x:x + w] = True
    level = float(left.max()) + 0.5          # one clipping level for all patches, above the texture
    left = np.where(flat, level, left)
    disp = gaussian_filter(rng.random((H, Wp)), 12.0)
    disp = (disp - disp.min()) / (disp.max() - disp.min() + 1e-12) * 0.8 * (D - 1)
    xs = np.arange(Wp)[None, :] + disp
    x0 = np.floor(xs).astype(int).clip(0, Wp - 1)
    x1 = (x0 + 1).clip(0, Wp - 1)
    f = xs - np.floor(xs)
    rows = np.arange(H)[:, None]
    right = left[rows, x0] * (1 - f) + left[rows, x1] * f
    rflat = flat[rows, x0] & flat[rows, x1]   # both interpolation taps inside a patch: the value is the level exactly
    right = np.where(rflat, level, right + noise * rng.standard_normal(right.shape))
    return normalize_image(left[:, :W]), normalize_image(right[:, :W])


def sample_pair(H=None, W=None):
    """The reference's one real input pair (samples/input/kittiL.png / kittiR.png, 370 x 1226, 8-bit grey; committed as
    tests/golden/kitti_sample_pair.npz by tests/golden/make_sample_pair.py), normalised as main.lua:1095-1096 does.
    Other sizes than 370 x 1226: the pair mirror-tiled (left / right and top / bottom reflections alternate, so no seams)
    and cropped -- a second pair with real-scene arm statistics at 1000 x 1500."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_sample_pair.npz"))
    l, r = z["left"].astype(np.float64), z["right"].astype(np.float64)
    if H is not None and (H, W) != l.shape:
        def tile(img):
            h, w = img.shape
            row = np.concatenate([img, img[:, ::-1]], 1)
            row = np.tile(row, (1, W // (2 * w) + 1))[:, :W]
            full = np.concatenate([row, row[::-1]], 0)
            return np.tile(full, (H // (2 * h) + 1, 1))[:H]
        l, r = tile(l), tile(r)
    return normalize_image(l), normalize_image(r)


def random_pair(H, W, seed=0):
    rng = np.random.default_rng(seed)
    return (normalize_image(rng.standard_normal((H, W))), normalize_image(rng.standard_normal((H, W))))


def blocky_pair(H, W, seed=0, levels=4):
    """Piecewise-constant images: long cross arms and many exact intensity ties."""
    rng = np.random.default_rng(seed)
    def one():
        small = rng.integers(0, levels, size=((H + 7) // 8, (W + 7) // 8)).astype(np.float32)
        return np.kron(small, np.ones((8, 8), np.float32))[:H, :W] * 0.25
    return one(), one()


def features(C, H, W, seed=42):
    """rng(seed) N(0,1) (2,C,H,W) L2-normalised over C (Normalize2 semantics, eps 1e-5)."""
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((2, C, H, W)).astype(np.float32)
    n = np.sqrt((f.astype(np.float64) ** 2).sum(1, keepdims=True) + 1e-5)
    return (f / n).astype(np.float32)


def raw_volumes(D, H, W, seed=7):
    """uniform[0,1) (sigmoid range) with the NaN triangles: left d > x, right x + d >= W."""
    rng = np.random.default_rng(seed)
    vl = rng.random((D, H, W), dtype=np.float32)
    vr = rng.random((D, H, W), dtype=np.float32)
    d = np.arange(D)[:, None, None]
    x = np.arange(W)[None, None, :]
    vl[np.broadcast_to(d > x, vl.shape)] = np.nan
    vr[np.broadcast_to(x + d >= W, vr.shape)] = np.nan
    return vl, vr


def same_bits(a, b):
    """Bit-exact equality with NaN == NaN (NaN masks must match exactly)."""
    a = np.ascontiguousarray(a, np.float32).ravel()
    b = np.ascontiguousarray(b, np.float32).ravel()
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    return np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32))


def diff_report(a, b, name=""):
    a = np.asarray(a, np.float32).ravel()
    b = np.asarray(b, np.float32).ravel()
    na, nb = np.isnan(a), np.isnan(b)
    bad_nan = int((na != nb).sum())
    ok = ~(na | nb)
    neq = ok & (a.view(np.uint32) != b.view(np.uint32))
    md = float(np.abs(a[ok] - b[ok]).max()) if ok.any() else 0.0
    first = np.flatnonzero(neq | (na != nb))[:5]
    return "%s: nan-mask mismatches=%d, value mismatches=%d / %d, max|diff|=%g, first idx=%s a=%s b=%s" % (
        name, bad_nan, int(neq.sum()), a.size, md, first.tolist(), a[first].tolist(), b[first].tolist())
