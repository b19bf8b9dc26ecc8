"""On the GPU box: the op-by-op route (mc.stereo_predict: what an unchanged main.lua drives through the shim, ~25 adcensus.* calls + tensor glue per
pair) and the fused entry (mc_predict) at KITTI-fast size, N pairs each after a warm-up and a pause -- for a rocprofv3 kernel trace
(scripts/rocpd_overlap.py splits busy time from host gaps).
    python scripts/gpu_ops_route.py ops|fused [N]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import mc_cnn_amd as mc  # noqa: E402
from mc_cnn_amd.predict import Workspace  # noqa: E402

which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = bench.CONFIGS["kitti_fast"]
preset, H, W, D, C, _ = cfg
prm = dict(mc.PRESETS[preset])
dev = torch.device("cuda", 0)
xb, kw, _ = bench.make_inputs(cfg, 0, dev, "sample")
ws = Workspace(prm, D, H, W, dev)
out = torch.empty((1, 1, H, W), dtype=torch.float32, device=dev)


def one():
    if which == "ops":
        return mc.stereo_predict(xb, prm, D, **kw)
    return mc.stereo_predict_fused(xb, prm, D, workspace=ws, out=out, **kw)


for _ in range(3):
    one()
torch.cuda.synchronize()
time.sleep(0.3)
t0 = time.perf_counter()
for _ in range(n):
    one()
torch.cuda.synchronize()
print("%s: %.3f ms per pair (host clock, %d pairs)" % (which, (time.perf_counter() - t0) / n * 1e3, n))
