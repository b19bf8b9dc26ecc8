"""cbca_lean2x_kernel adds the nine values of a minimal 3 x 3 support as a0 + a1 + ... + a8 -- eight additions -- where the reference
starts its accumulator at +0.0 and adds nine times (adcensus.cu:356-373), and takes the nine-addition chain only where its range test
on the sum (2^-95 <= |s| < 2^125: what the three-operation division by nine covers, tests/test_div9.py) fails.  Claim: wherever the
range test passes, both chains end in the same bits.  (0 + a0 differs from a0 only for a0 = -0.0 (+0.0 instead); a following addend
that is not a zero, or is +0.0, makes the partial sums equal again; so the sums differ only if all nine are -0.0 -- a zero, outside the
range.)  Checked here on float32 nine-tuples with zeros of both signs, denormals, huge values, infinities and NaNs mixed in."""
import numpy as np


def chains(a):
    """a: (N, 9) float32 -> (sum from +0.0 with nine additions, sum a0 + a1 + ... with eight), one rounding per addition"""
    with np.errstate(all="ignore"):
        nine = np.zeros(a.shape[0], np.float32)
        for k in range(9):
            nine = (nine + a[:, k]).astype(np.float32)
        eight = a[:, 0].copy()
        for k in range(1, 9):
            eight = (eight + a[:, k]).astype(np.float32)
    return nine, eight


def test_eight_additions_equal_nine_wherever_the_range_test_passes():
    rng = np.random.default_rng(5)
    n = 400000
    a = (rng.standard_normal((n, 9)) * np.exp(rng.uniform(-60, 60, (n, 1)))).astype(np.float32)
    special = np.array([0.0, -0.0, 1e-42, -1e-42, 3e38, -3e38, np.inf, -np.inf, np.nan, 1.0, -1.0], np.float32)
    mask = rng.random((n, 9)) < rng.choice([0.0, 0.2, 0.6, 1.0], (n, 1))
    a = np.where(mask, special[rng.integers(0, len(special), (n, 9))], a)
    a[:2000] = np.where(rng.random((2000, 9)) < 0.5, np.float32(-0.0), np.float32(0.0))   # nothing but zeros of both signs
    a[2000:2100] = np.float32(-0.0)
    nine, eight = chains(a)
    with np.errstate(all="ignore"):
        mag = np.abs(eight)
        in_range = (mag >= np.float32(2.0 ** -95)) & (mag < np.float32(2.0 ** 125))   # div9_ok(): false for NaN
    assert in_range.sum() > n // 4
    assert np.array_equal(nine[in_range].view(np.uint32), eight[in_range].view(np.uint32))
    # ... and the cases the range test exists for do occur in the sample: the two chains differ on all-(-0.0) tuples
    diff = nine.view(np.uint32) != eight.view(np.uint32)
    assert diff[2000:2100].all() and not (diff & in_range).any()


def test_the_kernel_source_says_so():
    import os
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mc-cnn_amd", "csrc", "cbca_lean.hip")).read()
    body = src[src.index("__device__ __forceinline__ cb_f4 lean_row("):src.index("template <int R>\n__device__ __forceinline__ L2xEntry") if "template <int R>\n__device__ __forceinline__ L2xEntry" in src else None]
    assert "float t = a.c[j] + a.c[j + 1];" in body and "fast = fast && div9_ok(t);" in body
    assert body.index("if (__any(!fast))") < body.index("float t = 0;")   # the exact path restarts the chain from +0.0
