"""GPU, under rocprofv3 --pmc: one classification, then lean passes at 1000x1500x256 on the textured pair in the launch variants given on the
command line (rb:variant ...), each N times -- scripts/gpu_lean_pmc.sh sums FETCH_SIZE / WRITE_SIZE per (kernel, grid)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib
mc = importlib.import_module("mc-cnn_amd")
from util import smooth_pair
A = mc.adcensus
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
H, W, D, L1, tau1 = 1000, 1500, 256, 14, 0.02
x0, x1 = smooth_pair(H, W, D, seed=1234)
xb = dev(np.stack([x0, x1]))[:, None]
x0c = torch.empty((1, 4, H, W), device="cuda"); x1c = torch.empty_like(x0c)
A.cross(xb[0:1], x0c, L1, tau1); A.cross(xb[1:2], x1c, L1, tau1)
vin = torch.rand((1, D, H, W), device="cuda")
out = torch.empty_like(vin)
for spec in sys.argv[1:]:
    rb, variant = (int(v) for v in spec.split(":"))
    A.cbca_cfg(x0c, x1c, vin, out, -1, form=8, rb=rb, d0=variant)
    for _ in range(4):
        A.cbca_cfg(x0c, x1c, vin, out, -1, form=9, rb=rb, d0=variant)
torch.cuda.synchronize()
