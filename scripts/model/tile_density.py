"""CPU model (tuning aid, no GPU): could cross-based aggregation be routed PER TILE instead of per pair (VERDICT r4 #1b)?
For a pair with real-scene arm statistics at 1000 x 1500 (L1 = 14, tau1 = 0.02): per wave tile of the texture route's two-pass kernel
(8 rows x 252 columns of one disparity plane) the number of outputs whose support is not the minimal 3 x 3 -- the entries its record
would have to hold (255 slots) -- as a histogram over the tiles, and the share of the plane that lies in tiles the texture route could
take (sparse: <= 64 entries, i.e. where its per-entry lanes stay a small part of the wave's work; possible: <= 255).
    python scripts/model/tile_density.py [natural|sample|mixed] [d ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cpu_oracle as oracle
from util import mixed_pair, natural_pair, sample_pair

H, W, D, L1, tau1 = 1000, 1500, 256, 14, 0.02
which = sys.argv[1] if len(sys.argv) > 1 else "natural"
ds = [int(a) for a in sys.argv[2:]] or [7, 60, 130, 200]
x0, x1 = {"natural": lambda: natural_pair(H, W, D, seed=1234), "mixed": lambda: mixed_pair(H, W, D, seed=1234), "sample": lambda: sample_pair(H, W)}[which]()
a0 = np.asarray(oracle.cross(x0, L1, tau1)).reshape(4, H, W).astype(np.int64)
a1 = np.asarray(oracle.cross(x1, L1, tau1)).reshape(4, H, W).astype(np.int64)
ys, xs = np.mgrid[0:H, 0:W]
dec = lambda a: (xs - a[0] - 1, a[1] - xs - 1, ys - a[2] - 1, a[3] - ys - 1)
A0, A1 = dec(a0), dec(a1)
TR, TC = 8, 252
edges = [0, 1, 8, 32, 64, 128, 255, 10 ** 9]
hist = np.zeros(len(edges) - 1, np.int64)
vox = np.zeros(len(edges) - 1, np.int64)
nonmin_total = out_total = 0
for d in ds:
    xp = np.clip(xs[0] - d, 0, W - 1)
    l, r, u, dn = (np.minimum(A0[k], A1[k][:, xp]) for k in range(4))
    unit_row = (l == 1) & (r == 1)
    up = np.vstack([unit_row[:1], unit_row[:-1]]); down = np.vstack([unit_row[1:], unit_row[-1:]])
    minimal = unit_row & (u == 1) & (dn == 1) & up & down
    partner = np.broadcast_to((xs[0] - d >= 0)[None, :], (H, W))
    listed = partner & ~minimal
    nonmin_total += int(listed.sum()); out_total += int(partner.sum())
    for ty in range(0, H, TR):
        for tx in range(0, W, TC):
            n = int(listed[ty:ty + TR, tx:tx + TC].sum())
            k = np.searchsorted(edges, n, side="right") - 1
            hist[k] += 1; vox[k] += min(TR, H - ty) * min(TC, W - tx)
print("%s pair, %dx%d, L1 = %d, tau1 = %g, planes d = %s: %.1f %% of the outputs with a partner have a support that is not the minimal 3 x 3" %
      (which, H, W, L1, tau1, ds, 100.0 * nonmin_total / out_total))
print("entries per wave tile (8 x 252 x 1)   tiles     share of tiles   share of the plane")
for k in range(len(hist)):
    hi = "%d" % (edges[k + 1] - 1) if edges[k + 1] < 10 ** 9 else "inf (record overflows)"
    print("  %4d .. %-24s %8d   %6.1f %%         %6.1f %%" % (edges[k], hi, hist[k], 100.0 * hist[k] / hist.sum(), 100.0 * vox[k] / vox.sum()))
print("tiles with <= 64 entries: %.1f %% of the plane; with <= 255 (the record holds them): %.1f %%" % (100.0 * vox[:4].sum() / vox.sum(), 100.0 * vox[:6].sum() / vox.sum()))
