"""CPU model (tuning aid, no GPU): the tile kernel's plan-reading pass as a discrete-event simulation of ONE block walking a strip --
today's barrier per step against a rolling scheme in which a mover wave commits the next step's rows as soon as the step before the
current one has completed, so that the block's compute waves run up to one step ahead instead of waiting at the barrier.
Chunk costs: the ISA-count model of tile_steps.py (plan-reading instance, with the three-row class).
    python scripts/model/tile_roll.py [natural|sample] [d ...] [--strips=N]
Output: utilisation of the compute waves (work / (waves x elapsed)) under both schemes and the instruction totals with the per-step
fixed work on all waves (today) or on the mover only."""
import os, sys, heapq
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cpu_oracle as oracle
from util import natural_pair, sample_pair

H, W, D, L1, tau1 = 1000, 1500, 256, 14, 0.02
A, TW = 13, 128
TH = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--th=")), 16)
which = sys.argv[1] if len(sys.argv) > 1 else "natural"
ds = [int(a) for a in sys.argv[2:] if not a.startswith("-")] or [60]
NSTRIPS = next((int(a.split('=')[1]) for a in sys.argv if a.startswith('--strips=')), 4)
x0, x1 = natural_pair(H, W, D, seed=1234) if which == "natural" else sample_pair(H, W)
def arms(x):
    a = np.asarray(oracle.cross(x, L1, tau1)).reshape(4, H, W).astype(np.int32)
    ys, xs = np.mgrid[0:H, 0:W]
    return xs - a[0] - 1, a[1] - xs - 1, ys - a[2] - 1, a[3] - ys - 1
l0, r0, u0, d0_ = arms(x0); l1, r1, u1, d1_ = arms(x1)
ROW_BASE, ROW_EV, TAP, SETUP, FINISH, FAST, FIXED = 14 + 10, 24, 5, 150, 60, 120, 250

def step_costs(n, u, dn, ok, ty, tx):
    """chunk costs (instructions) of the step at rows ty.., columns tx.., tallest first"""
    ys = np.arange(ty, min(H, ty + TH))
    th = len(ys)
    if th < TH: return []   # (ragged last step: ignored)
    okc = ok[tx:tx + TW]
    U = u[ty:ty + TH, tx:tx + TW]; Dn = dn[ty:ty + TH, tx:tx + TW]
    tw = U.shape[1]
    s0 = (ys[:, None] - U).reshape(TH // 4, 4, tw); e0 = (ys[:, None] + Dn).reshape(TH // 4, 4, tw)
    top = s0.min(1); bot = e0.max(1); ext = bot - top + 1
    mini = (U.reshape(TH // 4, 4, tw) == 1).all(1) & (Dn.reshape(TH // 4, 4, tw) == 1).all(1)
    items = []
    for g in range(TH // 4):
        for c in range(tw):
            if not okc[c]: continue
            t, e = top[g, c], ext[g, c]
            runs = n[max(t, 0):t + e, tx + c]
            is_mini = mini[g, c] and e == 6 and (runs == 3).all()
            reg3 = (not is_mini) and mini[g, c] and e == 6
            evrows = np.zeros(e, bool); evrows[s0[g, :, c] - t] = True; evrows[e0[g, :, c] - t] = True
            items.append((1 if is_mini else (2 if reg3 else e + 1), runs, evrows))
    items.sort(key=lambda it: -it[0])
    costs = []
    for i in range(0, len(items), 64):
        ch = items[i:i + 64]
        if ch[0][0] == 1: costs.append(FAST); continue
        if ch[0][0] == 2:
            c_ = 60
            for r in range(6):
                mx = max((it[1][r] if r < len(it[1]) else 0) for it in ch)
                c_ += 10 + 9 + (1 + (1, 2, 3, 3, 2, 1)[r]) * mx
            costs.append(c_); continue
        E = max(len(it[1]) for it in ch)
        c_ = SETUP + FINISH
        for r in range(E):
            mx = max((it[1][r] if r < len(it[1]) else 0) for it in ch)
            ev = any((r < len(it[2]) and it[2][r]) for it in ch)
            c_ += ROW_BASE + (ROW_EV if ev else 0) + TAP * mx + (9 if mx > 9 else 0) + (9 if mx > 18 else 0)
        costs.append(c_)
    return costs

def sim_barrier(steps, nw=8):
    t = 0.0
    for costs in steps:
        free = [0.0] * nw
        for c in costs:   # tallest first onto the first free wave
            f = heapq.heappop(free); heapq.heappush(free, f + c)
        t += max(free) + FIXED
    return t

def sim_roll(steps, nw=8, ahead=1, mover=120):
    """waves take chunks of step s in order; step s + ahead + 1 becomes ready `mover` after step s has completed"""
    ns = len(steps)
    ready = [0.0] * (ns + ahead + 2)
    done_t = [0.0] * ns
    free = [(0.0, w) for w in range(nw)]; heapq.heapify(free)
    for s, costs in enumerate(steps):
        fin = ready[s]
        for c in costs:
            f, w = heapq.heappop(free)
            st = max(f, ready[s])
            heapq.heappush(free, (st + c, w))
            fin = max(fin, st + c)
        done_t[s] = fin
        ready[s + ahead + 1] = max(ready[s + ahead], fin + mover)
    return max(done_t)

tot_work = tot_bar = tot_roll1 = tot_roll2 = 0.0; nsteps = 0
for d in ds:
    sh = -d
    xs = np.arange(W); ok = (xs + sh >= 0) & (xs + sh < W); xp = np.clip(xs + sh, 0, W - 1)
    l = np.minimum(l0, l1[:, xp]); r = np.minimum(r0, r1[:, xp]); u = np.minimum(u0, u1[:, xp]); dn = np.minimum(d0_, d1_[:, xp])
    n = l + r + 1
    for tx in list(range(0, W - TW + 1, TW))[::max(1, (W // TW) // NSTRIPS)][:NSTRIPS]:
        if not ok[tx:tx + TW].any(): continue
        steps = [step_costs(n, u, dn, ok, ty, tx) for ty in range(0, H - TH + 1, TH)]
        steps = [s for s in steps if s]
        work = sum(sum(s) for s in steps)
        tb, t1, t2 = sim_barrier(steps), sim_roll(steps, ahead=1), sim_roll(steps, ahead=2)
        print("d %3d strip x=%4d: %d steps, work %.0f K instr; barrier: %.0f K (util %.2f)  roll(1 ahead): %.0f K (util %.2f)  roll(2 ahead): %.0f K (util %.2f)" % (
            d, tx, len(steps), work / 1e3, tb / 1e3, work / (8 * tb), t1 / 1e3, work / (8 * t1), t2 / 1e3, work / (8 * t2)))
        tot_work += work; tot_bar += tb; tot_roll1 += t1; tot_roll2 += t2; nsteps += len(steps)
print("total: work %.0f K instr in %d steps (%.0f per step) + fixed: today %.0f per step on 8 waves, rolling ~150 per step on the mover" % (tot_work / 1e3, nsteps, tot_work / nsteps, 8 * FIXED))
print("utilisation of the 8 compute waves: barrier %.3f, rolling 1 ahead %.3f, 2 ahead %.3f" % (tot_work / (8 * tot_bar), tot_work / (8 * tot_roll1), tot_work / (8 * tot_roll2)))
print("instructions per step incl. fixed: today %.0f, rolling %.0f (%.2f x)" % (tot_work / nsteps + 8 * FIXED, tot_work / nsteps + 150, (tot_work / nsteps + 150) / (tot_work / nsteps + 8 * FIXED)))
