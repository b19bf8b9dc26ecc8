"""GPU: the list kernel on an all-flat pair (every support 27 x 27) -- time per tap of the largest size class."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib
mc = importlib.import_module("mc-cnn_amd")
A = mc.adcensus
H, W, D, L1, tau1 = 1000, 1500, 16, 14, 0.02
xb = torch.zeros((2, 1, H, W), device="cuda")
x0c = torch.empty((1, 4, H, W), device="cuda"); x1c = torch.empty_like(x0c)
A.cross(xb[0:1], x0c, L1, tau1); A.cross(xb[1:2], x1c, L1, tau1)
vin = torch.rand((1, D, H, W), device="cuda")
o = torch.empty_like(vin); o2 = torch.empty_like(vin)
for _ in range(2): A.cbca_cfg(x0c, x1c, vin, o, -1, form=3)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3): A.cbca_cfg(x0c, x1c, vin, o, -1, form=3)
torch.cuda.synchronize(); t = (time.time() - t0) / 3
A.cbca_reference_shaped(x0c, x1c, vin, o2, -1)
torch.cuda.synchronize(); t0 = time.time()
A.cbca_reference_shaped(x0c, x1c, vin, o2, -1)
torch.cuda.synchronize(); td = time.time() - t0
taps = D * H * W * 729.0
print("all-flat %dx%dx%d: listed %.2f ms (%.2f ps/tap), one thread per voxel %.2f ms" % (H, W, D, t * 1e3, t / taps * 1e12, td * 1e3))
from bench import same_bits_dev
print("same bits", same_bits_dev(o, o2))
