"""Tuning aid: time mc_fc_stack (accurate net's FC stack) at KITTI size and report TFLOP/s."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mc_cnn_amd as mc
from mc_cnn_amd.fc import fc_cost_volumes
H, W, D, C, NHID = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (370, 1226, 228, 112, 3))]
g = torch.Generator(device="cuda").manual_seed(1)
feat = torch.rand((2, C, H, W), device="cuda", generator=g)
dims = [2 * C] + [384] * (NHID + 1) + [1]
layers = [((torch.rand((dims[i + 1], dims[i]), device="cuda", generator=g) - 0.5) * (2 / dims[i] ** 0.5),
           (torch.rand((dims[i + 1],), device="cuda", generator=g) - 0.5) * 0.1) for i in range(len(dims) - 1)]
need = mc._lib.lib.mc_fc_stack_workspace_bytes(C, len(layers), H, W)
ws = torch.empty(need + 16, dtype=torch.uint8, device="cuda")
fc_cost_volumes(feat, layers, D, workspace=ws); torch.cuda.synchronize()
t0 = time.perf_counter()
fc_cost_volumes(feat, layers, D, workspace=ws); torch.cuda.synchronize()
dt = time.perf_counter() - t0
vox = sum(max(0, W - d) for d in range(D)) * H
flop = vox * 2.0 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
print("fc_stack %dx%dx%d C=%d hidden=%d: %.1f ms, %.1f TFLOP/s (reference-equivalent flops %.1f T), %.1f M voxels" %
      (H, W, D, C, NHID, dt * 1e3, flop / dt / 1e12, flop / 1e12, vox / 1e6))
