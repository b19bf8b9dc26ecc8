"""Tuning aid: SGM stage time on a few short images (per-step time of one wave's recurrence chain)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mc_cnn_amd as mc
from mc_cnn_amd.predict import Workspace
from util import features, smooth_pair
prm = dict(mc.PRESETS["kitti_fast"])
for (H, W, D) in [(1, 1226, 228), (8, 1226, 228), (64, 1226, 228), (370, 1226, 228)]:
    x0, x1 = smooth_pair(H, W, 8, seed=5)
    f = torch.from_numpy(features(16, H, W, seed=6)).cuda()
    xb = torch.from_numpy(np.stack([x0, x1])[:, None]).cuda()
    ws = Workspace(prm, D, H, W, xb.device)
    out = torch.empty((1, 1, H, W), device="cuda")
    for _ in range(3):
        r = mc.stereo_predict_fused(xb, prm, D, feat=f, workspace=ws, out=out, timed=True)
    print(os.environ.get("MC_SGM_FUSED", "1"), (H, W, D), "sgm ms", round(r["stage_ms"]["sgm"], 4), "per step us", round(r["stage_ms"]["sgm"] * 1e3 / (2 * W), 3) if os.environ.get("MC_SGM_FUSED", "1") != "0" else "")
