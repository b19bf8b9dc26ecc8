"""GPU debug: op-by-op route vs fused route on the tiled sample pair (mb_slow, D=64), several times: which outputs differ?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib
mc = importlib.import_module("mc-cnn_amd")
from util import sample_pair, raw_volumes
from bench import same_bits_dev
preset, H, W, D = sys.argv[1] if len(sys.argv) > 1 else "mb_slow", 1000, 1500, 64
if preset == "kitti_slow": H, W, D = 370, 1226, 228
prm = dict(mc.PRESETS[preset])
x0, x1 = sample_pair(H, W)
dev = torch.device("cuda", 0)
xb = torch.from_numpy(np.stack([x0, x1])[:, None]).to(dev)
vl, vr = raw_volumes(D, H, W, seed=11)
kw = dict(raw=(torch.from_numpy(vl).to(dev), torch.from_numpy(vr).to(dev)))
f = mc.stereo_predict_fused(xb, prm, D, want_volumes=True, want_disp0=True, **kw)
torch.cuda.synchronize()
for rep in range(4):
    o = mc.stereo_predict(xb, prm, D, return_all=True, **kw)
    torch.cuda.synchronize()
    msg = []
    for k in ("volL", "volR", "dispL0", "dispR0", "disp"):
        a, b = o[k].reshape(-1), f[k].reshape(-1)
        na, nb = torch.isnan(a), torch.isnan(b)
        bad = (na != nb) | (~na & (a.view(torch.int32) != b.view(torch.int32)))
        n = int(bad.sum())
        first = int(torch.nonzero(bad)[0]) if n else -1
        msg.append("%s:%d@%d" % (k, n, first))
    print("rep", rep, " ".join(msg), flush=True)
