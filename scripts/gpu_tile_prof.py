"""GPU (library built with -DMC_TILE_PROF): phase timeline (s_memtime, 100 MHz ticks) of one block of the cbca tile kernel's
plan-reading instance: per step and wave the time in the chunk phase, at the barrier behind it, in commit / stores / requests"""
import os, sys, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib
mc = importlib.import_module("mc-cnn_amd")
from util import natural_pair, smooth_pair
A = mc.adcensus
name = sys.argv[1] if len(sys.argv) > 1 else "natural"
H, W, D, L1, tau1 = (1000, 1500, 256, 14, 0.02) if "kitti" not in sys.argv else (370, 1226, 228, 5, 0.13)
x0, x1 = (smooth_pair if name == "smooth" else natural_pair)(H, W, D, seed=1234)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
xb = dev(np.stack([x0, x1]))[:, None]
x0c = torch.empty((1, 4, H, W), device="cuda"); x1c = torch.empty_like(x0c)
A.cross(xb[0:1], x0c, L1, tau1); A.cross(xb[1:2], x1c, L1, tau1)
vin = torch.rand((1, D, H, W), device="cuda"); o = torch.empty_like(vin)
A.cbca_cfg(x0c, x1c, vin, o, -1, form=4 if L1 <= 5 else 5)
for _ in range(2):
    A.cbca_cfg(x0c, x1c, vin, o, -1, form=6 if L1 <= 5 else 7)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "mc-cnn_amd", "libmcadcensus.so"))
buf = (ctypes.c_ulonglong * 4096)()
print("rc", lib.mc_debug_tile_prof(buf, 4096))
t = np.array(buf[:], dtype=np.int64).reshape(-1, 8, 16)
NW = 8 if L1 > 5 else 4
for step in range(1, 15):
    r = t[step, :NW]
    if r[0, 1] == 0: continue
    t0 = r[:, 1].min()
    chunks = r[:, 6] - r[:, 1]; b3 = r[:, 7] - r[:, 6]; commit = r[:, 8] - r[:, 7]; rest = r[:, 9] - r[:, 8]; b4 = r[:, 10] - r[:, 9]
    print("step %2d total %5d | chunk phase per wave %s | wait at its barrier %s | commit %d stores+requests %d barrier %d" % (
        step, int(r[:, 10].max() - t0), " ".join("%4d" % v for v in chunks), " ".join("%4d" % v for v in b3), int(commit.mean()), int(rest.mean()), int(b4.mean())))
