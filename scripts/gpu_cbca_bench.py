"""GPU: time single CBCA launches at the benchmarked sizes through the test hook (mc_cbca_ws_cfg): strip kernel (one
iteration) against the fused two-iteration kernel, rows per strip varied.  Arms from bench.py's synthetic pair."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import mc_cnn_amd as mc
import bench

def run(cfgname, variants):
    cfg = bench.CONFIGS[cfgname]
    preset, H, W, D, C, _ = cfg
    prm = mc.PRESETS[preset]
    xb, kw, _ = bench.make_inputs(cfg, 0, torch.device("cuda", 0))
    x0c = torch.empty((1, 4, H, W), device="cuda"); x1c = torch.empty((1, 4, H, W), device="cuda")
    mc.adcensus.cross(xb[0, 0].contiguous(), x0c, prm["L1"], prm["tau1"])
    mc.adcensus.cross(xb[1, 0].contiguous(), x1c, prm["L1"], prm["tau1"])
    vin = kw["raw"][0].reshape(1, D, H, W)
    a = torch.empty_like(vin); b = torch.empty_like(vin)
    V = 4.0 * D * H * W
    for name, kwargs in variants:
        its = 2 if kwargs.get("fused") == 1 else 1
        kwargs = dict(kwargs)
        mc.adcensus.cbca_cfg(x0c, x1c, vin, a, -1, **kwargs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 6
        e0.record()
        src, dst = a, b
        for _ in range(n):
            mc.adcensus.cbca_cfg(x0c, x1c, src, dst, -1, **kwargs)
            src, dst = dst, src
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print("%-10s %-28s %8.3f ms/launch  %8.3f ms/iteration  algorithmic %6.0f GB/s (%.3f of 8 TB/s)" % (
            cfgname, name, ms, ms / its, its * 2 * V / ms / 1e6, its * 2 * V / ms / 1e6 / 8000))

if __name__ == "__main__":
    which = sys.argv[1:] or ["mb_slow", "kitti_slow"]
    variants = [("warmup (first touch)", dict(fused=2)), ("lean", dict(fused=2)), ("lean rb=64", dict(fused=2, rb=64)),
                ("lean nt=0", dict(fused=2, nt=0)), ("fused2", dict(fused=1))]
    for c in which:
        run(c, variants)
