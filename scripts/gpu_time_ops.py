"""Tuning aid: time individual C-ABI operators at KITTI size with HIP events (torch.cuda.Event on the current stream)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mc_cnn_amd as mc
from util import smooth_pair
H, W, D = 370, 1226, 228
x0, x1 = smooth_pair(H, W, D, seed=1)
x0d, x1d = torch.from_numpy(x0).cuda(), torch.from_numpy(x1).cuda()
out = torch.empty((1, D, H, W), device="cuda")
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("ad      %.3f ms" % timeit(lambda: mc.adcensus.ad(x0d[None, None], x1d[None, None], out, -1)))
c0 = torch.stack([x0d, x0d * 0.5, x1d])[None].contiguous(); c1 = torch.stack([x1d, x1d * 0.5, x0d])[None].contiguous()
print("census1 %.3f ms" % timeit(lambda: mc.adcensus.census(x0d[None, None], x1d[None, None], out, -1)))
print("census3 %.3f ms" % timeit(lambda: mc.adcensus.census(c0, c1, out, -1)))
