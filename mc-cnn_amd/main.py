"""`main.lua -a predict` / `-a time` on MI355X: the host side of the reference's predict path in Python.

    python -m mc_cnn_amd.main kitti fast -a predict -net_fname NET -left L.png -right R.png -disp_max 70

mirrors `./main.lua kitti fast -a predict ...` (main.lua:10-32 flags, 1084-1105 action): loads the two images, converts
RGB to luma, normalises each to zero mean / unit (unbiased) std on the host, uploads (2,1,H,W), runs the feature net
(arch fast: l1 x [3x3 conv, pad 1, ReLU] with no ReLU after the last conv, then Normalize2 -- main.lua:727-746; the
convolutions are `mc_conv3x3`, a hand-written fp32-MFMA implicit GEMM of libmcadcensus.so, like everything after them), then the post-CNN
pipeline in one `mc_predict` call, and writes `left.bin`, `right.bin` (1,D,H,W) and `disp.bin` (1,1,H,W), raw float32,
with the reference's messages.  `-a time` is main.lua:1140-1167 (min of N runs on an uninitialised batch).

Architectures: fast, slow (feature net + `mc_fc_stack`, main.lua:958-983), ad and census (no net: `mc_ad` / `mc_census_ws`
volumes from the image pair, main.lua:932-942).

-net_fname: the reference's Torch7 `net_*.t7` (main.lua:892-902; read by `t7.py`), an `.npz` with arrays
w1,b1,...,w<l1>,b<l1> (w_i: (fm, in, 3, 3); arch slow also fw1,fb1,...), or `random:<seed>` for a seeded random net
(there is no network access for trained weights).  Hyper-parameter flags (-L1 -tau1 -cbca_i1 -cbca_i2 -pi1 -pi2 -sgm_i
-sgm_q1 -sgm_q2 -alpha1 -tau_so -blur_sigma -blur_t) default to main.lua's per-(dataset, arch) tables.
"""
import argparse
import sys
import time

import numpy as np

from .binio import write_bin
from .params import NET_SHAPES, SM_SKIP, SM_TERMINATE, TABLES


def rgb2y(img):
    """image.rgb2y: Y = 0.299 R + 0.587 G + 0.114 B on a (3,H,W) float tensor."""
    return (0.299 * img[0] + 0.587 * img[1] + 0.114 * img[2])[None]


def load_image(path):
    """image.load(path, nil, 'byte'):float() -> (C,H,W) float32 in 0..255."""
    from PIL import Image
    a = np.asarray(Image.open(path))
    if a.ndim == 2:
        a = a[None]
    else:
        a = np.transpose(a[:, :, :3], (2, 0, 1))
    return a.astype(np.float32)


def normalize(x):
    """x:add(-x:mean()):div(x:std()) -- torch's unbiased std, accumulations in double (main.lua:1095-1096)."""
    # in the reference's order: the mean is subtracted in float32 first, then the std of THAT tensor divides it
    y = (x - np.float32(x.astype(np.float64).mean())).astype(np.float32)
    return (y / np.float32(y.astype(np.float64).std(ddof=1))).astype(np.float32)


def parse(argv):
    if len(argv) < 2 or argv[0] not in ("kitti", "kitti2015", "mb") or argv[1] not in ("fast", "slow", "ad", "census"):
        raise SystemExit("usage: main.py {kitti|kitti2015|mb} {fast|slow|ad|census} -a {predict|time} [flags]  (main.lua:10-13)")
    dataset, arch = argv[0], argv[1]
    t = TABLES[(dataset, arch)]
    ap = argparse.ArgumentParser(prog="main.py %s %s" % (dataset, arch), prefix_chars="-")
    ap.add_argument("-a", default="predict", choices=["predict", "time"])
    ap.add_argument("-net_fname", default="random:42")
    ap.add_argument("-left", default="")
    ap.add_argument("-right", default="")
    ap.add_argument("-disp_max", type=int, default=228 if dataset != "mb" else 200)
    ap.add_argument("-gpu", type=int, default=1, help="1-based, as cutorch.setDevice (main.lua:16,342)")
    ap.add_argument("-tiny", action="store_true")
    for k in ("L1", "cbca_i1", "cbca_i2", "sgm_i"):
        ap.add_argument("-" + k, type=int, default=t[k])
    for k in ("tau1", "pi1", "pi2", "sgm_q1", "sgm_q2", "alpha1", "tau_so", "blur_sigma", "blur_t"):
        ap.add_argument("-" + k, type=float, default=t[k])
    ap.add_argument("-sm_terminate", default="", choices=sorted(SM_TERMINATE), help="main.lua:25")
    ap.add_argument("-sm_skip", default="", choices=sorted(SM_SKIP), help="main.lua:26")
    opt = ap.parse_args(argv[2:])
    prm = dict(t)
    prm["sm_terminate"], prm["sm_skip"] = opt.sm_terminate, opt.sm_skip   # make_params maps the stage names
    for k in ("L1", "cbca_i1", "cbca_i2", "sgm_i", "tau1", "pi1", "pi2", "sgm_q1", "sgm_q2", "alpha1", "tau_so", "blur_sigma",
              "blur_t"):
        prm[k] = getattr(opt, k)
    return dataset, arch, opt, prm


FC_SHAPES = {"kitti": (4, 384), "kitti2015": (4, 384), "mb": (3, 384)}  # (l2, nh2), main.lua:76-77, 123-124


def load_net(net_fname, dataset, arch, n_input_plane=1):
    """[(w, b)] of the feature net: from the reference's `.t7` (torch.save(..., 'ascii'), main.lua:587-600), an .npz, or
    seeded random (`random:<seed>`)."""
    l1, fm = NET_SHAPES[(dataset, arch)]
    if net_fname.endswith(".t7"):
        from . import t7
        layers = t7.load_reference_net(net_fname, arch)[0]
        if not layers:
            raise ValueError("%s: no SpatialConvolution modules found" % net_fname)
        return layers
    if net_fname.startswith("random:"):
        rng = np.random.default_rng(int(net_fname.split(":")[1]))
        layers = []
        for i in range(l1):
            cin = n_input_plane if i == 0 else fm
            bound = 1.0 / np.sqrt(cin * 9)  # nn.SpatialConvolution:reset() range
            layers.append((rng.uniform(-bound, bound, (fm, cin, 3, 3)).astype(np.float32),
                           rng.uniform(-bound, bound, (fm,)).astype(np.float32)))
        return layers
    z = np.load(net_fname)
    return [(z["w%d" % (i + 1)].astype(np.float32), z["b%d" % (i + 1)].astype(np.float32)) for i in range(l1)]


def load_fc(net_fname, dataset):
    """[(w (out,in), b (out))] of net_te2 (arch slow, main.lua:688-695): from the reference's `.t7`, an .npz (fw1,fb1,...)
    or seeded random."""
    l1, fm = NET_SHAPES[(dataset, "slow")]
    if net_fname.endswith(".t7"):
        from . import t7
        fc = t7.load_reference_net(net_fname, "slow")[1]
        if not fc:
            raise ValueError("%s: no SpatialConvolution1_fw modules found" % net_fname)
        return fc
    l2, nh2 = FC_SHAPES[dataset]
    dims = [2 * fm] + [nh2] * l2 + [1]
    if net_fname.startswith("random:"):
        rng = np.random.default_rng(int(net_fname.split(":")[1]) + 1)
        out = []
        for i in range(len(dims) - 1):
            bound = (1.0 if i < len(dims) - 2 else 6.0) / np.sqrt(dims[i])
            out.append((rng.uniform(-bound, bound, (dims[i + 1], dims[i])).astype(np.float32),
                        rng.uniform(-bound, bound, (dims[i + 1],)).astype(np.float32)))
        return out
    z = np.load(net_fname)
    return [(z["fw%d" % (i + 1)].astype(np.float32), z["fb%d" % (i + 1)].astype(np.float32)) for i in range(len(dims) - 1)]


def device_layers(layers, device):
    """[(w, b)] as float32 tensors on `device`: upload a net once, not once per call."""
    import torch
    return [tuple(t if isinstance(t, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32)).to(device) for t in wb)
            for wb in layers]


def features_slow(x_batch, layers):
    """forward_free(net_te, x_batch) for arch slow (main.lua:681-686): l1 x [3x3 conv, pad 1, ReLU]."""
    from . import adcensus
    h = x_batch.contiguous()
    for w, b in device_layers(layers, h.device):
        h = adcensus.conv3x3(h, w, b, relu=True)
    return h


def raw_volumes_slow(feat, fc_layers, disp_max, border_n):
    """main.lua:958-983: the FC stack for every (pixel, disparity) -> NaN-filled volumes, then fix_border."""
    import torch
    from . import adcensus
    from .fc import fc_cost_volumes
    dl = [(torch.from_numpy(w).to(feat.device), torch.from_numpy(b).to(feat.device)) for w, b in fc_layers]
    vl, vr = fc_cost_volumes(feat, dl, disp_max)
    adcensus.fix_border(vl, border_n, -1)
    adcensus.fix_border(vr, border_n, 1)
    return vl, vr


def features_fast(x_batch, layers):
    """forward_free(net_te, x_batch) for arch fast (main.lua:945): convs (pad 1) + ReLU between, then Normalize2."""
    import torch
    from . import adcensus
    h = x_batch.contiguous()
    layers = device_layers(layers, h.device)
    for i, (w, b) in enumerate(layers):
        h = adcensus.conv3x3(h, w, b, relu=i < len(layers) - 1)
    norm = torch.empty((h.shape[0], 1) + tuple(h.shape[2:]), dtype=torch.float32, device=h.device)
    out = torch.empty_like(h)
    adcensus.Normalize_forward(h, norm, out)   # Normalize2.lua:8-13 -> adcensus.cu:1310-1333
    return out


def main(argv=None):
    dataset, arch, opt, prm = parse(list(sys.argv[1:] if argv is None else argv))
    import torch
    from .predict import Workspace, stereo_predict_fused
    if arch not in ("fast", "slow", "ad", "census"):
        raise SystemExit("main.py: unknown architecture %r (main.lua:11: fast | slow | ad | census)" % arch)
    dev = torch.device("cuda", opt.gpu - 1)
    torch.cuda.set_device(dev)
    learned = arch in ("fast", "slow")
    layers = device_layers(load_net(opt.net_fname, dataset, arch), dev) if learned else []   # resident: uploaded once
    fc_layers = load_fc(opt.net_fname, dataset) if arch == "slow" else None
    prm["border_n"] = len(layers)  # (1 + l1*(3-1) - 1) / 2, main.lua:382-391,923

    def run(x_batch, D, workspace=None, want_volumes=False):
        if not learned:  # main.lua:932-942: hand-crafted costs straight from the image pair, no border fix
            from . import adcensus
            cost = adcensus.ad if arch == "ad" else adcensus.census
            H, W = x_batch.shape[2:]
            volL = torch.empty((1, D, H, W), dtype=torch.float32, device=x_batch.device)
            volR = torch.empty_like(volL)
            adcensus.fill_nan(volL)
            adcensus.fill_nan(volR)
            cost(x_batch[0:1], x_batch[1:2], volL, -1)
            cost(x_batch[1:2], x_batch[0:1], volR, 1)
            return stereo_predict_fused(x_batch, prm, D, raw=(volL, volR), workspace=workspace, want_volumes=want_volumes)
        if arch == "fast":
            return stereo_predict_fused(x_batch, prm, D, feat=features_fast(x_batch, layers), workspace=workspace,
                                        want_volumes=want_volumes)
        raw = raw_volumes_slow(features_slow(x_batch, layers), fc_layers, D, prm["border_n"])
        return stereo_predict_fused(x_batch, prm, D, raw=raw, workspace=workspace, want_volumes=want_volumes)
    if opt.a == "time":  # main.lua:1140-1167
        if dataset == "mb":
            prm["left_only"] = 1  # outside `-a predict` dataset mb runs direction -1 only (mb_directions, main.lua:953-955)
        H, W, D = (240, 320, 32) if opt.tiny else ((350, 1242, 228) if dataset != "mb" else (1000, 1500, 200))
        x_batch = torch.empty((2, 1, H, W), dtype=torch.float32, device=dev).normal_()
        ws = Workspace(prm, D, H, W, dev)
        best = float("inf")
        for _ in range(30 if arch == "fast" else 3):  # main.lua:1152
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(x_batch, D, workspace=ws)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(best)
        return 0
    x0, x1 = load_image(opt.left), load_image(opt.right)
    if x0.shape[0] == 3:
        assert x1.shape[0] == 3
        x0, x1 = rgb2y(x0), rgb2y(x1)
    D = opt.disp_max
    x_batch = torch.from_numpy(np.stack([normalize(x0), normalize(x1)])).to(dev)  # (2,1,H,W)
    res = run(x_batch, D, want_volumes=True)
    torch.cuda.synchronize()
    H, W = x_batch.shape[2:]
    for name, key in (("right", "volR"), ("left", "volL")):  # main.lua:954-955 writes right.bin first
        print("Writing %s.bin, %d x %d x %d x %d" % (name, 1, D, H, W))
        write_bin("%s.bin" % name, res[key].cpu().numpy())
    print("Writing disp.bin, %d x %d x %d x %d" % (1, 1, H, W))
    write_bin("disp.bin", res["disp"].cpu().numpy())
    return 0


if __name__ == "__main__":
    sys.exit(main())
