"""CPU model (tuning aid, no GPU), round 6: the tile kernel's instruction budget per voxel on real-scene arms, split into what a walk row costs before
its taps (run lookup, ring arithmetic: ROW_BASE), its first- / last-row events (ROW_EV), its taps (1 compare + K additions each, at the pace of the chunk's
longest run) and the per-chunk / per-step fixed parts -- for today's items (a column x K = 4 output rows) and for two candidates:
  desc   a per-(item, walk row) descriptor stream written by the plan pass (run address, length, reset / read-out bits in one or two dwords):
         ROW_BASE 24 -> 9 (descriptor load is a row ahead, one multiply-free address, one bit-field extract), ROW_EV 24 -> 16
  K8     items of a column x 8 output rows (16-row steps: two row groups): one compare per 8 additions and half the rows per output, at the price of
         additions into accumulators whose support does not hold the row (reset / read-out trick) and 8 compares + 16 selects per event row
    python scripts/model/tile_r6.py [natural|sample|mixed] [d ...]
Instruction counts are the ISA counts of cbca_tile.hip (scripts/model/tile_steps.py); the estimate is per voxel of the planes sampled."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cpu_oracle as oracle
from util import natural_pair, sample_pair, mixed_pair

H, W, D, L1, tau1 = 1000, 1500, 256, 14, 0.02
A, TW, TH = 13, 128, 16
which = sys.argv[1] if len(sys.argv) > 1 else "natural"
ds = [int(a) for a in sys.argv[2:] if not a.startswith("-")] or [60, 130]
x0, x1 = {"natural": lambda: natural_pair(H, W, D, seed=1234), "mixed": lambda: mixed_pair(H, W, D, seed=1234), "sample": lambda: sample_pair(H, W)}[which]()
a0 = np.asarray(oracle.cross(x0, L1, tau1)).reshape(4, H, W).astype(np.int32)
a1 = np.asarray(oracle.cross(x1, L1, tau1)).reshape(4, H, W).astype(np.int32)

def decode(a):
    ys, xs = np.mgrid[0:H, 0:W]
    return xs - a[0] - 1, a[1] - xs - 1, ys - a[2] - 1, a[3] - ys - 1

l0, r0, u0, d0_ = decode(a0); l1, r1, u1, d1_ = decode(a1)

def run(K, ROW_BASE, ROW_EV, EVK, label, SETUP=150, FINISH=60, FAST=120, FIXED=250):
    """K outputs per item.  A tap costs 1 + K instructions at the pace of the chunk's longest run on that row."""
    parts = dict(base=0.0, ev=0.0, taps=0.0, chunk=0.0, fixed=0.0, fast=0.0)
    vox = 0
    adds = slots = 0
    for d in ds:
        sh = -d
        xs = np.arange(W)
        ok = (xs + sh >= 0) & (xs + sh < W)
        xp = np.clip(xs + sh, 0, W - 1)
        l = np.minimum(l0, l1[:, xp]); r = np.minimum(r0, r1[:, xp]); u = np.minimum(u0, u1[:, xp]); dn = np.minimum(d0_, d1_[:, xp])
        n = l + r + 1
        for ty in range(0, H - TH + 1, TH * 3):
            for tx in range(0, W - TW + 1, TW):
                ys = np.arange(ty, ty + TH)
                okc = ok[tx:tx + TW]
                U = u[ty:ty + TH, tx:tx + TW]; Dn = dn[ty:ty + TH, tx:tx + TW]
                G = TH // K
                s0 = (ys[:, None] - U).reshape(G, K, TW); e0 = (ys[:, None] + Dn).reshape(G, K, TW)
                top = s0.min(1); bot = e0.max(1)
                ext = bot - top + 1
                mini = (U.reshape(G, K, TW) == 1).all(1) & (Dn.reshape(G, K, TW) == 1).all(1)
                items = []
                for g in range(G):
                    for c in range(TW):
                        if not okc[c]: continue
                        t, e = top[g, c], ext[g, c]
                        runs = n[t:t + e, tx + c]
                        is_mini = mini[g, c] and (runs == 3).all()
                        evrows = np.zeros(e, bool)
                        evrows[s0[g, :, c] - t] = True; evrows[e0[g, :, c] - t] = True
                        useful = sum(n[s0[g, j, c]:e0[g, j, c] + 1, tx + c].sum() for j in range(K))
                        items.append((1 if is_mini else e + 1, runs, evrows, useful))
                vox += TW * TH
                if not items: continue
                items.sort(key=lambda it: -it[0])
                for i in range(0, len(items), 64):
                    ch = items[i:i + 64]
                    if ch[0][0] == 1:
                        parts["fast"] += FAST * K / 4; continue
                    E = max(len(it[1]) for it in ch)
                    parts["chunk"] += SETUP + FINISH * K / 4
                    for rr in range(E):
                        mx = max((it[1][rr] if rr < len(it[1]) else 0) for it in ch)
                        ev = any((rr < len(it[2]) and it[2][rr]) for it in ch)
                        parts["base"] += ROW_BASE
                        parts["ev"] += (ROW_EV * EVK) if ev else 0
                        parts["taps"] += (1 + K) * mx + (9 if mx > 9 else 0) + (9 if mx > 18 else 0)
                        slots += mx * 64 * K
                adds += sum(it[3] for it in items)
                parts["fixed"] += 8 * FIXED
    tot = sum(parts.values())
    print("%-34s %5.2f instr / voxel  =  " % (label, tot / vox) + "  ".join("%s %.2f" % (k, v / vox) for k, v in parts.items()) +
          "   | additions %.1f / voxel, of the add slots issued %.0f %% useful" % (adds / vox, 100.0 * adds / max(1, slots)))
    return tot / vox

print("%s pair, 1000 x 1500, L1 = 14, planes %s (every third step row)" % (which, ds))
b = run(4, 24, 24, 1.0, "today (K = 4)")
d_ = run(4, 9, 16, 1.0, "desc (K = 4, descriptor stream)")
k8 = run(8, 24, 24, 2.0, "K8 (items of 8 rows)")
k8d = run(8, 9, 16, 2.0, "K8 + desc")
k16 = run(16, 24, 24, 4.0, "K16 (items of 16 rows)")
print("relative to today: desc %.2f, K8 %.2f, K8 + desc %.2f, K16 %.2f" % (d_ / b, k8 / b, k8d / b, k16 / b))
