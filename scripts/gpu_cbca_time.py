"""GPU: single-launch timings of the CBCA kernels at a benchmarked size through the test hooks, under rocprofv3 --stats."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mc_cnn_amd as mc
import bench
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "mb_slow"]
preset, H, W, D, C, _ = cfg
prm = dict(mc.PRESETS[preset])
xb, kw, _ = bench.make_inputs(cfg, 0, torch.device("cuda", 0))
x0c = torch.empty((1, 4, H, W), device="cuda"); x1c = torch.empty((1, 4, H, W), device="cuda")
mc.adcensus.cross(xb[0, 0].contiguous(), x0c, prm["L1"], prm["tau1"])
mc.adcensus.cross(xb[1, 0].contiguous(), x1c, prm["L1"], prm["tau1"])
vin = kw["raw"][0].reshape(1, D, H, W)
a = torch.empty_like(vin); b = torch.empty_like(vin)
mc.adcensus.cbca_fused2(x0c, x1c, vin, a, -1); mc.adcensus.cbca_fused2(x0c, x1c, a, b, -1)
for _ in range(4):
    mc.adcensus.cbca_fused2(x0c, x1c, a, b, -1)
    mc.adcensus.cbca_fused2(x0c, x1c, a, b, -1, rb=100)
    mc.adcensus.cbca_cfg(x0c, x1c, a, b, -1)
torch.cuda.synchronize()
