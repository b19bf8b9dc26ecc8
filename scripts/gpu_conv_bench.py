"""On the GPU box: mc_conv3x3 against torch's convolution (MIOpen) on the feature nets' layer shapes, fp32.
    python scripts/gpu_conv_bench.py [--no-torch] [--reps 20]
Prints one line per shape: ms and TFLOP/s (2*N*H*W*Cin*Cout*9 flops) of both."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mc_cnn_amd as mc  # noqa: E402

SHAPES = [(2, 1, 64, 370, 1226), (2, 64, 64, 370, 1226), (2, 112, 112, 370, 1226), (2, 64, 64, 1000, 1500), (2, 112, 112, 1000, 1500)]


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-torch", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--shape", type=str, default="", help="N,Cin,Cout,H,W instead of the layer shapes")
    a = ap.parse_args()
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.benchmark = True
    shapes = [tuple(int(v) for v in a.shape.split(","))] if a.shape else SHAPES
    for si, (N, Cin, Cout, H, W) in enumerate(shapes):
        if a.only >= 0 and si != a.only:
            continue
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn((N, Cin, H, W), device="cuda", generator=g)
        bound = 1.0 / np.sqrt(Cin * 9)
        w = (torch.rand((Cout, Cin, 3, 3), device="cuda", generator=g) * 2 - 1) * bound
        b = (torch.rand((Cout,), device="cuda", generator=g) * 2 - 1) * bound
        out = torch.empty((N, Cout, H, W), device="cuda")
        flops = 2.0 * N * H * W * Cin * Cout * 9
        t_mc = timed(lambda: mc.adcensus.conv3x3(x, w, b, True, out=out), a.reps)
        line = "N=%d %3d->%3d %dx%d: mc_conv3x3 %.3f ms (%.1f TFLOP/s)" % (N, Cin, Cout, H, W, t_mc, flops / t_mc / 1e9)
        if not a.no_torch:
            import torch.nn.functional as F
            t_t = timed(lambda: F.relu_(F.conv2d(x, w, b, padding=1)), a.reps)
            ref = F.relu_(F.conv2d(x, w, b, padding=1))
            err = float((ref - out).abs().max())
            line += ", torch/MIOpen conv+relu %.3f ms (%.1f TFLOP/s), max|diff| %.2e" % (t_t, flops / t_t / 1e9, err)
        print(line, flush=True)


if __name__ == "__main__":
    main()
