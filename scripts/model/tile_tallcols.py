"""CPU model (tuning aid, no GPU): what would 'tall column units' on v_mfma_f32_16x16x4_f32 save the tile kernel (DESIGN section 7, round 5 lead)?
The step's columns whose four items are ALL tall (>= KT rows) leave the item table and are walked as units of 16 columns x 16 outputs: one MFMA adds
four run values to the 16 accumulators of each of 16 columns.  Instruction estimates per unit from the ISA of the existing walk (row overhead, events)
and 6 instructions per group of four taps; everything else as scripts/model/tile_steps.py counts it.
    python scripts/model/tile_tallcols.py [natural|sample|mixed] [d ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cpu_oracle as oracle
from util import mixed_pair, natural_pair, sample_pair

H, W, D, L1, tau1 = 1000, 1500, 256, 14, 0.02
A, TW, TH, KT = 13, 128, 16, 14
which = sys.argv[1] if len(sys.argv) > 1 else "natural"
ds = [int(a) for a in sys.argv[2:] if not a.startswith("-")] or [7, 60, 130, 200]
x0, x1 = {"natural": lambda: natural_pair(H, W, D, seed=1234), "mixed": lambda: mixed_pair(H, W, D, seed=1234), "sample": lambda: sample_pair(H, W)}[which]()
a0 = np.asarray(oracle.cross(x0, L1, tau1)).reshape(4, H, W).astype(np.int64)
a1 = np.asarray(oracle.cross(x1, L1, tau1)).reshape(4, H, W).astype(np.int64)
ys_, xs_ = np.mgrid[0:H, 0:W]
dec = lambda a: (xs_ - a[0] - 1, a[1] - xs_ - 1, ys_ - a[2] - 1, a[3] - ys_ - 1)
A0, A1 = dec(a0), dec(a1)
ROW_BASE, ROW_EV, TAP, SETUP, FINISH, FAST, FIXED = 24, 24, 5, 150, 60, 120, 250
U_ROW, U_EV, U_GROUP, U_SETUP = 16, 24, 6, 260     # the unit's row: slot / address / run word, events (asm form), per group of four taps; set-up + 4 divisions + stores
tot = dict(old=0.0, new=0.0, vox=0, tallcols=0, cols=0, unit_work=0.0, old_tall_work=0.0)
for d in ds:
    xp = np.clip(xs_[0] - d, 0, W - 1)
    ok = (xs_[0] - d >= 0)
    l, r, u, dn = (np.minimum(A0[k], A1[k][:, xp]) for k in range(4))
    n = l + r + 1
    for ty in range(0, H - TH + 1, TH * 3):
        for tx in range(0, W - TW + 1, TW):
            U = u[ty:ty + TH, tx:tx + TW]; Dn = dn[ty:ty + TH, tx:tx + TW]
            ys = np.arange(ty, ty + TH)
            s0 = (ys[:, None] - U).reshape(4, 4, TW); e0 = (ys[:, None] + Dn).reshape(4, 4, TW)
            top = s0.min(1); bot = e0.max(1); ext = bot - top + 1          # (g, c)
            okc = ok[tx:tx + TW]
            tallcol = (ext >= KT).all(0) & okc
            def chunk_cost(items):   # items: list of (ext, runs array)
                items.sort(key=lambda it: -it[0])
                c = 0.0
                for i in range(0, len(items), 64):
                    ch = items[i:i + 64]
                    E = ch[0][0]
                    if E <= 6 and all(it[0] == 6 and (it[1] == 3).all() for it in ch):
                        c += FAST; continue
                    cc = SETUP + FINISH
                    for row in range(E):
                        mx = max((it[1][row] if row < len(it[1]) else 0) for it in ch)
                        cc += ROW_BASE + ROW_EV + TAP * mx + (9 if mx > 9 else 0) + (9 if mx > 18 else 0)
                    c += cc
                return c
            allitems, rest, tallitems = [], [], []
            for g in range(4):
                for c in range(TW):
                    if not okc[c]: continue
                    it = (int(ext[g, c]), n[top[g, c]:top[g, c] + ext[g, c], tx + c])
                    allitems.append(it)
                    (tallitems if tallcol[c] else rest).append(it)
            old = chunk_cost(list(allitems)) + 8 * FIXED
            new = chunk_cost(list(rest)) + 8 * FIXED
            cols = np.flatnonzero(tallcol)
            # units of 16 tall columns, in column order (neighbouring columns of a flat region have similar supports)
            for i in range(0, len(cols), 16):
                cc = cols[i:i + 16]
                ctop = top[:, cc].min(0); cbot = bot[:, cc].max(0)
                E = int((cbot - ctop + 1).max())
                w = U_SETUP
                for row in range(E):
                    mx = 0
                    for k, c in enumerate(cc):
                        y = ctop[k] + row
                        if y <= cbot[k]: mx = max(mx, int(n[y, tx + c]))
                    w += U_ROW + U_EV + U_GROUP * ((mx + 3) // 4)
                new += w; tot["unit_work"] += w
            tot["old_tall_work"] += chunk_cost(list(tallitems)) if tallitems else 0
            tot["old"] += old; tot["new"] += new; tot["vox"] += TW * TH
            tot["tallcols"] += int(tallcol.sum()); tot["cols"] += int(okc.sum())
print("%s pair, planes %s: tall columns (all four items >= %d rows) %.1f %% of the columns" % (which, ds, KT, 100.0 * tot["tallcols"] / tot["cols"]))
print("instructions per voxel (model): today %.2f, with tall-column units %.2f  (%.1f %% fewer); the tall columns' items cost %.2f per voxel today, as units %.2f" % (
    tot["old"] / tot["vox"], tot["new"] / tot["vox"], 100.0 * (1 - tot["new"] / tot["old"]), tot["old_tall_work"] / tot["vox"], tot["unit_work"] / tot["vox"]))
