"""GPU: cbca on a pair with real-scene arm statistics (tests/util.natural_pair) at KITTI size, kitti-slow thresholds."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib
mc = importlib.import_module("mc-cnn_amd")
from util import natural_pair, smooth_pair
from bench import same_bits_dev
A = mc.adcensus
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
import itertools
cases = list(itertools.product(((370, 1226, 228, 5, 0.13), (1000, 1500, 256, 14, 0.02)), (("smooth", smooth_pair), ("natural", natural_pair))))
if "--mb-natural" in sys.argv:
    cases = cases[3:]
for (H, W, D, L1, tau1), (name, mk) in cases:
    x0, x1 = mk(H, W, D, seed=1234)
    xb = dev(np.stack([x0, x1]))[:, None]
    x0c = torch.empty((1, 4, H, W), device="cuda"); x1c = torch.empty_like(x0c)
    A.cross(xb[0:1], x0c, L1, tau1); A.cross(xb[1:2], x1c, L1, tau1)
    vin = torch.rand((1, D, H, W), device="cuda")
    o1 = torch.empty_like(vin); o2 = torch.empty_like(vin)
    o3 = torch.empty_like(vin)
    for fn, tag in ((lambda: A.cbca(x0c, x1c, vin, o1, -1), "strip"), (lambda: A.cbca_reference_shaped(x0c, x1c, vin, o2, -1), "direct"),
                    (lambda: A.cbca_cfg(x0c, x1c, vin, o3, -1, form=2 if L1 <= 5 else 3), "window" if L1 <= 5 else "listed (incl. classification)")):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        print(H, L1, name, tag, "ms/call", round((time.time() - t0) * 200, 3), flush=True)
    print(name, "same bits", same_bits_dev(o1, o2), same_bits_dev(o3, o2), flush=True)
