"""-m gpu: the accurate net's FC stack (mc_fc_stack, fp32 MFMA) against the oracle's restatement of main.lua:958-983.
Bar: NaN masks identical; values within 1e-4 (north_star's float tolerance) -- the reference's own GEMM order is
cuBLAS's and unpinned, so this operator is not held to bit-exactness."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def make_layers(C, n_hidden, seed):
    rng = np.random.default_rng(seed)
    dims = [2 * C] + [384] * (n_hidden + 1) + [1]
    layers = []
    for i in range(len(dims) - 1):
        bound = (1.0 if i < len(dims) - 2 else 6.0) / np.sqrt(dims[i])  # wider output layer: outputs spread over (0,1)
        layers.append((rng.uniform(-bound, bound, (dims[i + 1], dims[i])).astype(np.float32),
                       rng.uniform(-bound, bound, (dims[i + 1],)).astype(np.float32)))
    return layers


@pytest.mark.parametrize("C,H,W,D,n_hidden", [(8, 5, 50, 10, 3), (16, 3, 100, 7, 2), (112, 2, 40, 48, 3), (4, 9, 97, 5, 1)])
def test_fc_stack(mc, oracle, C, H, W, D, n_hidden):
    from util import features
    from mc_cnn_amd.fc import fc_cost_volumes
    f = features(C, H, W, seed=C + W)
    f = np.maximum(f, 0) * 3.0  # post-ReLU-like, non-negative features as net_te produces
    layers = make_layers(C, n_hidden, seed=W)
    want_l, want_r = oracle.fc_stack(f[0], f[1], D, layers)
    dl = [(torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda()) for w, b in layers]
    vl, vr = fc_cost_volumes(torch.from_numpy(f).cuda(), dl, D)
    torch.cuda.synchronize()
    for got, want, name in ((vl, want_l, "left"), (vr, want_r, "right")):
        g = got.cpu().numpy()[0]
        assert np.array_equal(np.isnan(g), np.isnan(want)), "%s: NaN mask differs" % name
        ok = ~np.isnan(want)
        err = np.abs(g[ok] - want[ok]).max()
        assert err <= 1e-4, "%s: max |diff| = %g" % (name, err)
        assert want[ok].std() > 1e-3, want[ok].std()  # the comparison is not vacuous (outputs are not saturated)


@pytest.mark.parametrize("C,H,W,D,n_hidden", [(112, 3, 300, 20, 3), (32, 4, 250, 9, 2)])
def test_fc_stack_vs_torch_addmm_chain(mc, C, H, W, D, n_hidden):
    """Second, independent reference: what the reference's own modules do (SpatialConvolution1_fw.lua:11-31: `addmm` of the
    (out,in) weight with the (in, pixels) activations + bias, nn.ReLU between, nn.Sigmoid at the end; main.lua:958-983
    builds the input by concatenating featL[:,y,x] and featR[:,y,x-d]) as a plain fp32 torch addmm chain per disparity.
    Widths straddle several 96-voxel tiles of the MFMA kernel.  Tolerance 1e-4: BLAS summation order is third-party."""
    from util import features
    from mc_cnn_amd.fc import fc_cost_volumes
    torch.backends.cuda.matmul.allow_tf32 = False
    f = np.maximum(features(C, H, W, seed=C + W), 0) * 3.0
    layers = make_layers(C, n_hidden, seed=W + 1)
    ft = torch.from_numpy(f).cuda()
    dl = [(torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda()) for w, b in layers]
    vl, vr = fc_cost_volumes(ft, dl, D)
    want_l = torch.full((D, H, W), float("nan"), device="cuda")
    want_r = torch.full((D, H, W), float("nan"), device="cuda")
    for d in range(D):
        n = W - d
        if n <= 0:
            break
        x = torch.cat([ft[0][:, :, d:], ft[1][:, :, :n]], 0).reshape(2 * C, H * n)      # (in, pixels), main.lua:968-972
        for li, (w, b) in enumerate(dl):
            x = torch.addmm(b[:, None], w, x)                                               # SpatialConvolution1_fw.lua:21
            x = torch.relu(x) if li < len(dl) - 1 else torch.sigmoid(x)
        s = x.reshape(H, n)
        want_l[d, :, d:] = s
        want_r[d, :, :n] = s
    torch.cuda.synchronize()
    for got, want, name in ((vl[0], want_l, "left"), (vr[0], want_r, "right")):
        assert torch.equal(torch.isnan(got), torch.isnan(want)), "%s: NaN mask differs" % name
        ok = ~torch.isnan(want)
        err = float((got[ok] - want[ok]).abs().max())
        assert err <= 1e-4, "%s: max |diff| = %g" % (name, err)
        assert float(want[ok].std()) > 1e-3


def test_main_predict_arch_slow(mc, oracle, tmp_path, monkeypatch):
    """`main.py kitti slow -a predict ...`: conv features -> FC stack -> fix_border -> stereo_predict.  The FC stack is
    held to 1e-4; everything after it is bit-exact given the same raw volumes, which is what is checked here: the
    .bin files against the oracle pipeline run on the raw volumes the device produced."""
    from PIL import Image
    from scipy.ndimage import gaussian_filter
    from mc_cnn_amd import main as mcmain
    from util import diff_report, same_bits
    H, W, D = 24, 80, 12
    rng = np.random.default_rng(5)
    base = gaussian_filter(rng.random((H, W + 6)), 2.0)
    base = ((base - base.min()) / np.ptp(base) * 255).astype(np.uint8)
    Image.fromarray(base[:, 6:]).save(tmp_path / "L.png")
    Image.fromarray(base[:, :W]).save(tmp_path / "R.png")
    monkeypatch.chdir(tmp_path)
    assert mcmain.main(["kitti", "slow", "-a", "predict", "-net_fname", "random:3", "-left", "L.png", "-right", "R.png",
                        "-disp_max", str(D)]) == 0
    disp = mc.read_bin("disp.bin", (1, 1, H, W))
    left = mc.read_bin("left.bin", (1, D, H, W))
    x0 = mcmain.normalize(mcmain.load_image("L.png"))
    x1 = mcmain.normalize(mcmain.load_image("R.png"))
    xb = torch.from_numpy(np.stack([x0, x1])).cuda()
    layers = mcmain.load_net("random:3", "kitti", "slow")
    fcl = mcmain.load_fc("random:3", "kitti")
    feat = mcmain.features_slow(xb, layers)
    vl, vr = mcmain.raw_volumes_slow(feat, fcl, D, len(layers))
    # FC stack vs oracle restatement (tolerance)
    wl, wr = oracle.fc_stack(feat[0].cpu().numpy(), feat[1].cpu().numpy(), D, fcl)
    wl = oracle.fix_border(wl, len(layers), -1)
    ok = ~np.isnan(wl)
    assert np.array_equal(np.isnan(vl.cpu().numpy()[0]), np.isnan(wl))
    assert np.abs(vl.cpu().numpy()[0][ok] - wl[ok]).max() <= 1e-4
    # everything downstream: bit-exact on the device's raw volumes
    prm = dict(mc.TABLES[("kitti", "slow")])
    prm["border_n"] = len(layers)
    want = oracle.stereo_predict(prm, x0[0], x1[0], D, rawL=vl.cpu().numpy()[0], rawR=vr.cpu().numpy()[0])
    assert same_bits(left, want["volL"]), diff_report(left, want["volL"], "left.bin")
    assert same_bits(disp, want["disp"]), diff_report(disp, want["disp"], "disp.bin")
