"""GPU (library built with -DMC_TILE_PROF): phase timeline (s_memtime, 100 MHz ticks) of one block of the cbca tile kernel"""
import os, sys, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib
mc = importlib.import_module("mc-cnn_amd")
from util import natural_pair, smooth_pair
A = mc.adcensus
name = sys.argv[1] if len(sys.argv) > 1 else "smooth"
H, W, D, L1, tau1 = (1000, 1500, 256, 14, 0.02) if "kitti" not in sys.argv else (370, 1226, 228, 5, 0.13)
x0, x1 = (smooth_pair if name == "smooth" else natural_pair)(H, W, D, seed=1234)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
xb = dev(np.stack([x0, x1]))[:, None]
x0c = torch.empty((1, 4, H, W), device="cuda"); x1c = torch.empty_like(x0c)
A.cross(xb[0:1], x0c, L1, tau1); A.cross(xb[1:2], x1c, L1, tau1)
vin = torch.rand((1, D, H, W), device="cuda"); o = torch.empty_like(vin)
for _ in range(2):
    A.cbca_cfg(x0c, x1c, vin, o, -1, form=2 if L1 <= 5 else 3)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "mc-cnn_amd", "libmcadcensus.so"))
buf = (ctypes.c_ulonglong * 4096)()
print("rc", lib.mc_debug_tile_prof(buf, 4096))
t = np.array(buf[:], dtype=np.int64).reshape(-1, 4, 16)
names = ["fetch", "pass1", "B1", "scan+scatter", "B2", "chunks", "B3", "output", "commit", "B4"]
for step in range(2, 10):
    for w in (0, 3):
        r = t[step, w]
        if r[0] == 0: continue
        d = np.diff(r[:11])
        print("step", step, "wave", w, "total", int(r[10] - r[0]), " ".join("%s=%d" % (n, v) for n, v in zip(names, d)))
