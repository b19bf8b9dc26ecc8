"""CPU model (tuning aid, no GPU): what the tile kernel's steps look like on a pair -- per 128 x 16 step the items' heights, the
64-item chunks after the sort, an instruction estimate of every chunk (ISA counts of cbca_tile.hip) -- to see how much of a launch is
bound by the step's tallest chunk (one wave walking a chain) and how much by total instruction issue.
    python scripts/model/tile_steps.py [natural|sample] [d ...] [--reg3] [--wb=N] [--uni=E] [--size=HxWxD]
--reg3: with the class for items of four three-row outputs; --wb: secondary sort key (widest run // N); --uni: count rows of chunks of >= E rows on which
every walking lane has the same run length."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cpu_oracle as oracle
from util import natural_pair, sample_pair

H, W, D, L1, tau1 = 1000, 1500, 256, 14, 0.02
for a in sys.argv:
    if a.startswith("--size="):
        H, W, D = (int(v) for v in a.split("=")[1].split("x"))
A, TW, TH = 13, 128, 16
which = sys.argv[1] if len(sys.argv) > 1 else "natural"
REG3 = "--reg3" in sys.argv
UNI = next((int(a.split('=')[1]) for a in sys.argv if a.startswith('--uni=')), 99)
WB = next((int(a.split('=')[1]) for a in sys.argv if a.startswith('--wb=')), 0)   # secondary key: widest run // WB
ds = [int(a) for a in sys.argv[2:] if not a.startswith("-")] or [7, 60, 130, 200]
x0, x1 = natural_pair(H, W, D, seed=1234) if which == "natural" else sample_pair(H, W)
a0, a1 = oracle.cross(x0, L1, tau1), oracle.cross(x1, L1, tau1)   # (4,H,W): left, right, up, down  (adcensus.cu:280-322)
a0 = np.asarray(a0).reshape(4, H, W).astype(np.int32); a1 = np.asarray(a1).reshape(4, H, W).astype(np.int32)

def decode(a):
    # cross() stores the first coordinate that is NOT in the arm (adcensus.cu:318): xl, xr, yu, yd -> lengths
    ys, xs = np.mgrid[0:H, 0:W]
    return xs - a[0] - 1, a[1] - xs - 1, ys - a[2] - 1, a[3] - ys - 1

l0, r0, u0, d0_ = decode(a0); l1, r1, u1, d1_ = decode(a1)
print("arms ok:", l0.min(), l0.max(), u0.max())

ROW_BASE, ROW_EV, TAP, SETUP, FINISH, FAST, FIXED = 14 + 10, 24, 5, 150, 60, 120, 250   # instructions (VALU + LDS reads of a row's first nine)
tot = dict(steps=0, path=0.0, work=0.0, tall_steps=0, path_tall=0.0, adds=0, slots=0)
hist = np.zeros(40)
share = {}
def acc(k, v): share[k] = share.get(k, 0) + v
for d in ds:
    sh = -d
    xs = np.arange(W)
    ok = (xs + sh >= 0) & (xs + sh < W)
    xp = np.clip(xs + sh, 0, W - 1)
    l = np.minimum(l0, l1[:, xp]); r = np.minimum(r0, r1[:, xp]); u = np.minimum(u0, u1[:, xp]); dn = np.minimum(d0_, d1_[:, xp])
    n = l + r + 1
    for ty in range(0, H - TH + 1, TH * 3):          # every third step row (sampling)
        for tx in range(0, W - TW + 1, TW):
            ys = np.arange(ty, ty + TH)
            cols = np.arange(tx, tx + TW)
            okc = ok[cols]
            U = u[ty:ty + TH, tx:tx + TW]; Dn = dn[ty:ty + TH, tx:tx + TW]
            s0 = (ys[:, None] - U).reshape(TH // 4, 4, TW); e0 = (ys[:, None] + Dn).reshape(TH // 4, 4, TW)
            top = s0.min(1); bot = e0.max(1)                  # (g, c)
            ext = bot - top + 1
            mini = (U.reshape(TH // 4, 4, TW) == 1).all(1) & (Dn.reshape(TH // 4, 4, TW) == 1).all(1)
            # runs of the rows each item walks
            items = []
            for g in range(TH // 4):
                for c in range(TW):
                    if not okc[c]: continue
                    t, e = top[g, c], ext[g, c]
                    runs = n[t:t + e, tx + c]
                    is_mini = mini[g, c] and (runs == 3).all() and e == 6
                    evrows = np.zeros(e, bool)
                    evrows[s0[g, :, c] - t] = True; evrows[e0[g, :, c] - t] = True
                    reg3 = (not is_mini) and mini[g, c] and e == 6
                    full = (U.reshape(TH // 4, 4, TW)[g, :, c] == A).all() and (Dn.reshape(TH // 4, 4, TW)[g, :, c] == A).all()
                    acc("n_items", 1); acc("n_items_full", int(full)); acc("n_items_tall", int(e >= 20))
                    key = 1 if is_mini else (2 if (reg3 and REG3) else e + 1)
                    items.append((key, runs, evrows, sum(n[s0[g, j, c]:e0[g, j, c] + 1, tx + c].sum() for j in range(4)), reg3))
            if not items: continue
            items.sort(key=(lambda it: (-it[0], -(int(it[1].max()) // WB))) if WB else (lambda it: -it[0]))
            chunks = [items[i:i + 64] for i in range(0, len(items), 64)]
            costs = []
            for ch in chunks:
                if ch[0][0] == 1:
                    costs.append(FAST); acc("fast", FAST); continue
                if ch[0][0] == 2:   # all four outputs three rows tall: static rows, no events (row r feeds outputs max(0, r-2) .. min(3, r))
                    c_ = 60
                    for i in range(6):
                        mx = max(it[1][i] for it in ch)
                        c_ += 10 + 9 + (1 + (1, 2, 3, 3, 2, 1)[i]) * mx
                    costs.append(c_); acc("reg3", c_); continue
                E = max(len(it[1]) for it in ch)
                c_ = SETUP + FINISH
                for i in range(E):
                    mx = max((it[1][i] if i < len(it[1]) else 0) for it in ch)
                    if E >= UNI:
                        act = [it[1][i] for it in ch if i < len(it[1])]
                        acc("n_tall_rows", 1)
                        if min(act) == mx:
                            acc("n_tall_rows_uniform", 1); acc("n_uniform_cmpx_saved", mx)
                    ev = any((i < len(it[2]) and it[2][i]) for it in ch)
                    c_ += ROW_BASE + (ROW_EV if ev else 0) + TAP * mx + (9 if mx > 9 else 0) + (9 if mx > 18 else 0)
                    tot["slots"] += mx * 256
                costs.append(c_)
                acc("E<=6" if E <= 6 else "E7-12" if E <= 12 else "E13-20" if E <= 20 else "E21+", c_)
                acc("n_reg3_in_general", sum(1 for it in ch if it[4]))
                hist[min(39, E)] += 1
            tot["adds"] += sum(it[3] for it in items)
            path = max(costs) + FIXED
            work = sum(costs) + 8 * FIXED
            tot["steps"] += 1; tot["path"] += path; tot["work"] += work
            if max(costs) > 2000: tot["tall_steps"] += 1; tot["path_tall"] += path
s = tot["steps"]
print("steps %d: mean critical path %.0f instr, mean work %.0f instr (%.1f x path); steps with a chunk > 2000 instr: %.0f %% (their mean path %.0f)" % (
    s, tot["path"] / s, tot["work"] / s, tot["work"] / tot["path"], 100.0 * tot["tall_steps"] / s, tot["path_tall"] / max(1, tot["tall_steps"])))
print("adds per voxel %.1f, lockstep efficiency %.2f" % (tot["adds"] / (s * TW * TH), tot["adds"] / max(1, tot["slots"])))
# time model per step of one block: issue-bound  work * 2.85 * 3 / 4 cycles  (3 blocks share 4 SIMDs) vs chain-bound  path * 8.6 cycles
acc("fixed", 8 * FIXED * s)
tw = sum(v for k, v in share.items() if not k.startswith("n_"))
print("work share:", "  ".join("%s %.1f%%" % (k, 100.0 * v / tw) for k, v in share.items() if not k.startswith("n_")), " items of three-row outputs in general chunks:", share.get("n_reg3_in_general", 0))
print({k: v for k, v in share.items() if k.startswith("n_")})
print("chunk heights:", " ".join("%d:%d" % (i, h) for i, h in enumerate(hist) if h))
