"""On the GPU box: mc_conv3x3 on random shapes against a float64 torch convolution (tolerance 1e-4 of the output scale, as tests/test_gpu_conv.py).
    python scripts/gpu_conv_fuzz.py [cases] [seed]
Shapes are drawn so that every path is hit: one / several output-channel groups, tiles of 4 and of 8 rows, the chunked bank (Cin > 140), runs that
cross columns and images, launches with fewer units than waves and with many tiles per wave, odd channel counts, images narrower than a strip."""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mc_cnn_amd as mc  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
bad = 0
t0 = time.time()
for k in range(cases):
    kind = rng.integers(0, 6)
    N = int(rng.integers(1, 4))
    Cin = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 64, 100, 112, 141, 150, 200]))
    Cout = int(rng.choice([1, 5, 8, 24, 32, 40, 64, 64, 96, 100, 112, 128]))
    if kind == 0:
        H, W = int(rng.integers(1, 12)), int(rng.integers(1, 40))
    elif kind == 1:
        H, W = int(rng.integers(1, 80)), int(rng.integers(1, 200))
    elif kind == 2:
        H, W = int(rng.integers(100, 400)), int(rng.integers(20, 140))
    elif kind == 3:
        H, W = int(rng.integers(2, 30)), int(rng.integers(300, 1300))
    else:
        H, W = int(rng.integers(30, 160)), int(rng.integers(30, 300))
    if Cin * Cout * H * W * N > 6e9 / 18:   # keep a case under a few hundred ms of float64 reference
        Cin, Cout = min(Cin, 64), min(Cout, 64)
    relu = bool(rng.integers(0, 2))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((N, Cin, H, W), device="cuda", generator=g)
    bound = 1.0 / np.sqrt(Cin * 9)
    w = (torch.rand((Cout, Cin, 3, 3), device="cuda", generator=g) * 2 - 1) * bound
    b = (torch.rand((Cout,), device="cuda", generator=g) * 2 - 1) * bound
    got = mc.adcensus.conv3x3(x, w, b, relu)
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if relu:
        want = F.relu(want)
    torch.cuda.synchronize()
    err = float((got.double() - want).abs().max())
    scale = max(1.0, float(want.abs().max()))
    ok = err <= 1e-4 * scale and bool(torch.isfinite(got).all())
    if not ok:
        bad += 1
        print("MISMATCH case %d: N=%d %d->%d %dx%d relu=%s err %.3g scale %.3g" % (k, N, Cin, Cout, H, W, relu, err, scale), flush=True)
print("conv fuzz: %d cases, %d mismatching, %.0f s (seed %d)" % (cases, bad, time.time() - t0, seed))
sys.exit(1 if bad else 0)
