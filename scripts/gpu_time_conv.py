"""GPU: the hand-written 3x3 convolution (mc_conv3x3) against torch's (MIOpen) at the feature-net sizes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
import mc_cnn_amd as mc
torch.backends.cudnn.allow_tf32 = False
for (N, Cin, Cout, H, W) in [(2, 1, 64, 370, 1226), (2, 64, 64, 370, 1226), (2, 112, 112, 370, 1226), (2, 64, 64, 1000, 1500), (2, 112, 112, 1000, 1500)]:
    x = torch.randn((N, Cin, H, W), device="cuda"); w = torch.randn((Cout, Cin, 3, 3), device="cuda") * 0.05; b = torch.randn((Cout,), device="cuda")
    out = torch.empty((N, Cout, H, W), device="cuda")
    def t(f, n=5):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    ours = t(lambda: mc.adcensus.conv3x3(x, w, b, True, out=out))
    ref = t(lambda: F.relu(F.conv2d(x, w, b, padding=1)))
    fl = 2.0 * 9 * Cin * Cout * H * W * N
    print("N=%d %3d->%3d %dx%d: mc_conv3x3 %.3f ms (%.1f TFLOP/s), torch/MIOpen conv+relu %.3f ms (%.1f TFLOP/s)" % (N, Cin, Cout, H, W, ours, fl / ours / 1e9, ref, fl / ref / 1e9))
