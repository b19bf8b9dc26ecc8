"""-m gpu: the hand-written 3x3 convolution of the feature net (mc_conv3x3, fp32 MFMA implicit GEMM, filter bank resident in LDS) against a plain
fp32 torch convolution (cudnn.SpatialConvolution(n_in, fm, 3, 3, 1, 1, 1, 1) + ReLU, main.lua:681-686, 727-746).
Tolerance 1e-4 relative to the output scale: the reference's cuDNN algorithm (and summation order) is chosen at run
time (cudnn.benchmark = true, main.lua:330), so this operator has no bit-exact target."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("N,Cin,Cout,H,W,relu", [
    (2, 1, 64, 37, 70, True),        # first layer of the nets (one luminance plane)
    (2, 64, 64, 33, 65, True),       # arch fast, inner layer
    (2, 64, 64, 20, 31, False),      # arch fast, last layer: no ReLU before Normalize2
    (2, 112, 112, 18, 45, True),     # arch slow
    (1, 3, 7, 9, 11, True),          # odd everything
    (2, 16, 128, 8, 40, False),      # two groups of 64 output channels
    (1, 40, 96, 5, 33, True),        # three groups of 32
    (1, 150, 40, 9, 70, True),       # a bank that does not fit even for 32 output channels: chunks of input channels, LDS reloaded
    (1, 64, 64, 1, 5, True),         # one row, less than a strip
    (3, 8, 64, 61, 33, False),       # runs that cross columns and images, tiles of 1 .. 4 rows
    (1, 64, 64, 300, 100, True),     # every wave of the launch busy, several tiles per wave, short last tiles
    (1, 112, 112, 130, 70, True),    # the same for the four-group split (tiles of 8 rows)
])
def test_conv3x3_vs_torch(mc, N, Cin, Cout, H, W, relu):
    import torch.nn.functional as F
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    rng = np.random.default_rng(Cin * 1000 + Cout)
    x = torch.from_numpy(rng.standard_normal((N, Cin, H, W)).astype(np.float32)).cuda()
    bound = 1.0 / np.sqrt(Cin * 9)
    w = torch.from_numpy(rng.uniform(-bound, bound, (Cout, Cin, 3, 3)).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.uniform(-bound, bound, (Cout,)).astype(np.float32)).cuda()
    got = mc.adcensus.conv3x3(x, w, b, relu)
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1)   # float64 reference: both fp32 orders are within tolerance of it
    if relu:
        want = F.relu(want)
    torch.cuda.synchronize()
    err = float((got.double() - want).abs().max())
    scale = float(want.abs().max())
    assert err <= 1e-4 * max(1.0, scale), "max |diff| = %g (scale %g)" % (err, scale)
    assert float(got.std()) > 1e-3


def test_feature_nets_run_on_the_hand_written_convolution(mc):
    """main.py's features_fast / features_slow no longer call torch's convolution"""
    import inspect
    from mc_cnn_amd import main as mcmain
    src = inspect.getsource(mcmain.features_fast) + inspect.getsource(mcmain.features_slow)
    assert "conv2d" not in src and "adcensus.conv3x3" in src
    layers = mcmain.load_net("random:3", "kitti", "fast")
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((2, 1, 24, 50)).astype(np.float32)).cuda()
    f = mcmain.features_fast(x, layers)
    assert f.shape == (2, 64, 24, 50)
    n = (f.double() ** 2).sum(1)
    assert float((n - 1).abs().max()) < 1e-3     # Normalize2: unit-norm feature vectors (eps 1e-5 inside the root)
