"""CPU: hand-computed micro-cases derived line by line from the cited reference kernels.  They pin the
oracle's conventions independently of the golden vectors (which pin its values): sign and shift of
StereoJoin (adcensus.cu:1455-1477; the stale test.lua:45-73 has the opposite sign), exclusive arm ends
and the skipped distance-1 test of `cross` (280-322), the intersected support of `cbca` (343-377), the
sgm2 recurrence with its NaN behaviour (535-618), arg-min ties / NaN (244-262), fix_border's Lua
negative indices (main.lua:922-927), and the .bin layout (samples/load_bin.py)."""
import numpy as np

from util import same_bits

NAN = np.nan


def f32(a):
    return np.asarray(a, np.float32)


def eq(got, want):
    assert same_bits(got, f32(want)), "\n got=%s\nwant=%s" % (np.asarray(got), np.asarray(want))


def test_stereo_join_sign_and_shift(oracle):
    # C=2, H=1, W=3, D=2: s(x,d) = -(L[0,x]*R[0,x-d] + L[1,x]*R[1,x-d]); volL[d,x]=s, volR[d,x-d]=s
    L = f32([[[1, 2, 3]], [[1, 1, 1]]])
    R = f32([[[4, 5, 6]], [[2, 2, 2]]])
    vl, vr = oracle.stereo_join(L, R, 2)
    eq(vl, [[[-6, -12, -20]], [[NAN, -10, -17]]])
    eq(vr, [[[-6, -12, -20]], [[-10, -17, NAN]]])


def test_cross_exclusive_ends_and_skipped_first_test(oracle):
    # one row [0,0,0,1,1]; L1=3, tau1=0.5.  Layout (4,H,W): -x, +x, -y, +y exclusive ends.
    img = f32([[0, 0, 0, 1, 1]])
    a = oracle.cross(img, 3, 0.5)
    eq(a[0], [[-1, -1, -1, 1, 2]])   # -x: x=3: xx=2 skipped (distance 1), xx=1: |1-0| >= tau -> stop at 1
    eq(a[1], [[3, 3, 4, 5, 5]])      # +x: x=0: xx=3 has |0-1| >= tau -> 3; x=2: xx=3 skipped, xx=4: |0-1| >= tau -> 4
    eq(a[2], [[-1] * 5])
    eq(a[3], [[1] * 5])


def test_cross_plus_x_detail(oracle):
    # the +x arm of x=2 above, spelled out: xx=3 (distance 1) is never tested, xx=4 fails the colour test
    a = oracle.cross(f32([[0, 0, 0, 1, 1]]), 3, 0.5)
    assert a[1, 0, 2] == 4.0


def test_cross_length_limit(oracle):
    # constant row, L1=3: the walk stops AT distance L1 (tested after the colour rule), so the arm
    # covers distances 1..L1-1 and the stored end is x+L1 (or the image border)
    a = oracle.cross(np.zeros((1, 8), np.float32), 3, 0.5)
    eq(a[1], [[3, 4, 5, 6, 7, 8, 8, 8]])
    eq(a[0], [[-1, -1, -1, 0, 1, 2, 3, 4]])
    # L1 = 0, tau1 = 0 (all fast presets): |c-c| >= 0 stops at distance 2 -> 3x3 minimum support
    a = oracle.cross(np.zeros((1, 8), np.float32), 0, 0.0)
    eq(a[1], [[2, 3, 4, 5, 6, 7, 8, 8]])


def test_cbca_region_mean(oracle):
    img = np.zeros((3, 3), np.float32)
    arms = oracle.cross(img, 0, 0.0)          # every arm ends at distance 2 or the border
    vol = np.zeros((2, 3, 3), np.float32)
    vol[0] = np.arange(1, 10).reshape(3, 3)
    vol[1] = np.arange(11, 20).reshape(3, 3)
    vol[1, :, 0] = NAN                          # left volume: d=1 invalid at x=0
    out = oracle.cbca(arms, arms, vol, -1)
    assert out[0, 1, 1] == np.float32(45.0 / 9.0)           # whole 3x3
    assert out[0, 0, 0] == np.float32((1 + 2 + 4 + 5) / 4.0)  # rows 0..1, cols 0..1
    assert np.isnan(out[1, :, 0]).all()                      # x + d*direction outside: copied through
    # (d=1, y=1, x=1): rows 0..2; cols = (max(-1, -1+1), min(3, 2+1)) = {1, 2} of plane 1
    want = np.float32(0)
    for v in (12, 13, 15, 16, 18, 19):
        want = np.float32(want + np.float32(v))
    assert out[1, 1, 1] == np.float32(want / np.float32(6))


def test_sgm2_recurrence_with_nan(oracle):
    # H=1, W=2, D=2, direction=-1, flat images; derivation in the test body comments
    x = np.zeros((1, 2), np.float32)
    C = f32([[[1, NAN], [3, 2]]])   # (H,W,D): left volume has d=1 invalid at x=0
    out = oracle.sgm2(x, x, C, 1.0, 4.0, 0.5, 2.0, 2.0, 2.0, -1)
    # ->: p0 = C; p1: prev=[1,NaN], m=1; d=0: P=(1,4): min(1, 5, NaN+1)=1 -> (3+1)-1 = 3
    #                                   d=1: partner's predecessor off-row -> D2=10 -> P=(.5,2): min(NaN,3,1.5)=1.5 -> 2.5
    # <-: p1 = C; p0: prev=[3,2], m=2; d=0: min(3, 6, 2+1)=3 -> (1+3)-2 = 2 ; d=1: (NaN+2)-2 = NaN
    # down, up: H=1, every pixel is a first pixel: L = C
    eq(out, [[[1 + 2 + 1 + 1, NAN], [3 + 3 + 3 + 3, 2.5 + 2 + 2 + 2]]])


def test_argmin_ties_and_nan(oracle):
    vol = f32([[[2, NAN, 5, NAN]], [[1, 3, 5, NAN]], [[1, NAN, 4, NAN]]])  # (D=3,H=1,W=4)
    eq(oracle.argmin(vol), [[1, 1, 2, 0]])  # first strict minimum; NaN never wins; all-NaN -> 0


def test_fix_border_lua_negative_indices(oracle):
    vol = np.arange(1 * 1 * 7, dtype=np.float32).reshape(1, 1, 7)
    # direction -1 (left volume): columns -1,-2 (from the right) <- column -(n+1) = 7-3 = index 4
    eq(oracle.fix_border(vol.copy(), 2, -1), [[[0, 1, 2, 3, 4, 4, 4]]])
    # direction +1 (right volume): columns 1,2 (1-based) <- column n+1 = index 2
    eq(oracle.fix_border(vol.copy(), 2, 1), [[[2, 2, 2, 3, 4, 5, 6]]])


def test_outlier_detection_labels(oracle):
    # x - d0 < 0 -> 1 ; |d0 - d1[x-d0]| < 1.1 -> 0 ; else 2 if some d matches d1[x-d], else 1
    d0 = f32([[3, 0, 1, 2]])
    d1 = f32([[0, 0, 5, 5]])
    # x=3: d0=2 -> d1[1]=0, |2-0| >= 1.1; scan d: |0-d1[3]|=5, |1-d1[2]|=4, |2-d1[1]|=2, |3-d1[0]|=3 -> no match -> 1
    eq(oracle.outlier_detection(d0, d1, 4), [[1, 0, 0, 1]])
    # with d1[2] = 1 the scan finds d=1: |1 - d1[3-1]| = 0 -> mismatch (2)
    d1b = f32([[0, 0, 1, 5]])
    eq(oracle.outlier_detection(d0, d1b, 4), [[1, 0, 0, 2]])


def test_bin_layout_roundtrip(tmp_path):
    """left.bin is raw little-endian float32 (1,D,H,W), C order, no header (samples/load_bin.py:3-5)."""
    import mc_cnn_amd as mc
    a = np.arange(2 * 3 * 4, dtype=np.float32).reshape(1, 2, 3, 4)
    a[0, 1, 2, 3] = np.nan
    p = str(tmp_path / "left.bin")
    mc.write_bin(p, a)
    raw = open(p, "rb").read()
    assert len(raw) == a.size * 4
    assert np.frombuffer(raw, "<f4")[1 * 12 + 0 * 4 + 1] == a[0, 1, 0, 1]  # d*H*W + y*W + x
    assert same_bits(np.memmap(p, dtype=np.float32, shape=(1, 2, 3, 4)), a)  # exactly load_bin.py's call
    assert same_bits(mc.read_bin(p, (1, 2, 3, 4)), a)
