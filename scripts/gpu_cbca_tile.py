"""GPU: the cbca kernels per regime -- strip / window / listed / one thread per voxel / tile kernel -- ms per call and bit
comparison, on the Gaussian texture and on the pair with real-scene arm statistics (tests/util)."""
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
only = [a for a in sys.argv[1:] if not a.startswith("-")]
only_tile = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--only-tile=")]
for (H, W, D, L1, tau1), (name, mk) in cases:
    if only and ("%d%s" % (L1, name)) not in only:
        continue
    x0, x1 = mk(H, W, D, seed=1234)
    xb = dev(np.stack([x0, x1]))[:, None]
    x0c = torch.empty((1, 4, H, W), device="cuda"); x1c = torch.empty_like(x0c)
    A.cross(xb[0:1], x0c, L1, tau1); A.cross(xb[1:2], x1c, L1, tau1)
    vin = torch.rand((1, D, H, W), device="cuda")
    ref = torch.empty_like(vin)
    A.cbca_reference_shaped(x0c, x1c, vin, ref, -1)
    forms = ((0, "adcensus.cbca", 0), (1, "strip", 0), (2, "tile<4> v0", 0), (2, "tile<4> v1", 1), (2, "tile<4> v2", 2), (2, "tile<4> v3", 3)) if L1 <= 5 else (
        (0, "adcensus.cbca", 0), (1, "strip", 0), (3, "tile<13> v0", 0), (3, "tile<13> v1", 1), (3, "tile<13> v2", 2), (3, "tile<13> v3", 3))
    forms = forms + (((4, "tile<4> sorts + writes the plan", 0), (6, "tile<4> reads the plan", 0)) if L1 <= 5 else
                     ((5, "tile<13> sorts + writes the plan", 0), (7, "tile<13> reads the plan", 0)))
    if "--plan-only" in sys.argv:
        forms = [f for f in forms if f[0] >= 4]
    if only_tile:
        forms = [f for f in forms if f[0] >= 2 and f[2] in only_tile]
    for form, tag, rb in forms:
        o = torch.full_like(vin, -7.0)
        fn = lambda: A.cbca_cfg(x0c, x1c, vin, o, -1, form=form, rb=rb)
        reps = 1 if "--once" in sys.argv else 5
        for _ in range(1 if "--once" in sys.argv else 2): fn()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        print(H, L1, name, tag, "ms/call (incl. pack)", round((time.time() - t0) * 1000 / reps, 3), "same bits", same_bits_dev(o, ref), flush=True)
