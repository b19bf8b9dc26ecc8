"""EXPERIMENT driver (GPU): deferred-queue cbca kernel vs the product strip kernel -- bits, then time."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib
mc = importlib.import_module("mc-cnn_amd")
from util import blocky_pair, random_pair, raw_volumes, smooth_pair
from bench import same_bits_dev

A = mc.adcensus
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def crosses(x0, x1, L1, tau1):
    H, W = x0.shape
    xb = dev(np.stack([x0, x1]))[:, None]
    x0c = torch.empty((1, 4, H, W), device="cuda"); x1c = torch.empty_like(x0c)
    A.cross(xb[0:1], x0c, L1, tau1); A.cross(xb[1:2], x1c, L1, tau1)
    return x0c, x1c


def special(vol, seed):
    rng = np.random.default_rng(seed)
    v = vol.copy().reshape(-1)
    n = v.size
    for val, frac in ((np.nan, 0.01), (0.0, 0.05), (-0.0, 0.05), (np.inf, 0.002), (1e-40, 0.01), (3e38, 0.002), (-3e38, 0.002)):
        idx = rng.integers(0, n, max(1, int(n * frac)))
        v[idx] = val
    return v.reshape(vol.shape)


bad = 0
cases = []
for (H, W, D) in [(90, 300, 9), (41, 519, 6), (27, 253, 5), (83, 64, 12), (130, 1000, 5), (64, 248, 4), (50, 249, 3), (33, 497, 4)]:
    for mk, L1, tau1 in (("smooth", 14, 0.02), ("random", 5, 0.13), ("blocky", 14, 0.2), ("smooth", 34, 0.03), ("random", 3, 0.5), ("smooth", 2, 0.02)):
        cases.append((H, W, D, mk, L1, tau1))
for ci, (H, W, D, mk, L1, tau1) in enumerate(cases):
    x0, x1 = {"smooth": lambda: smooth_pair(H, W, 8, seed=H + ci), "random": lambda: random_pair(H, W, seed=W + ci),
              "blocky": lambda: blocky_pair(H, W, seed=D + ci)}[mk]()
    x0c, x1c = crosses(x0, x1, L1, tau1)
    vl, vr = raw_volumes(D, H, W, seed=13 + ci)
    for sp in (0, 1):
        for direction, vol in ((-1, vl), (1, vr)):
            v = dev(special(vol, ci) if sp else vol)[None]
            want = torch.full((1, D, H, W), -7.0, device="cuda"); got = torch.full((1, D, H, W), -9.0, device="cuda")
            A.cbca(x0c, x1c, v, want, direction)
            for rb in (0, 16, 40):
                got.fill_(-9.0)
                A.cbca_dq(x0c, x1c, v, got, direction, rb=rb, nt=ci & 1)
                torch.cuda.synchronize()
                if not same_bits_dev(got, want):
                    bad += 1
                    if bad > 12: sys.exit(1)
                    g, w = got.cpu().numpy().reshape(-1), want.cpu().numpy().reshape(-1)
                    neq = ~((g.view(np.int32) == w.view(np.int32)) | (np.isnan(g) & np.isnan(w)))
                    i = np.flatnonzero(neq)
                    print("MISMATCH", (H, W, D, mk, L1, tau1), "sp", sp, "dir", direction, "rb", rb, "n", i.size,
                          [(int(k // (H * W)), int(k % (H * W) // W), int(k % W), float(g[k]), float(w[k])) for k in i[:4]], flush=True)
print("cases", len(cases), "mismatching runs", bad, flush=True)

if "--time" in sys.argv:
    for name, (H, W, D, L1, tau1) in (("mb", (1000, 1500, 256, 14, 0.02)), ("kitti", (370, 1226, 228, 5, 0.13))):
        x0, x1 = smooth_pair(H, W, D, seed=1234)
        x0c, x1c = crosses(x0, x1, L1, tau1)
        vin = torch.rand((1, D, H, W), device="cuda")
        o1 = torch.empty_like(vin); o2 = torch.empty_like(vin)
        for fn, out, tag in ((lambda: A.cbca(x0c, x1c, vin, o1, -1), o1, "strip"), (lambda: A.cbca_dq(x0c, x1c, vin, o2, -1), o2, "dq")):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(10): fn()
            torch.cuda.synchronize()
            print(name, tag, "ms/call incl. pack (+classify)", (time.time() - t0) * 100, flush=True)
        print(name, "same bits", same_bits_dev(o1, o2), flush=True)
