#!/usr/bin/env python
"""tests/golden/make_golden.py -- generates tests/golden/*.npz ON THE GPU BOX from the REFERENCE ITSELF
(oracle/_ref/libadcensus_ref.so = /root/reference/adcensus.cu compiled for gfx950, oracle/build_ref.py).

    gpurun -- 'python tests/golden/make_golden.py gpurun_out/golden'   # then copy *.npz to tests/golden/

Every file holds the inputs AND the reference's outputs, so the CPU tests (tests/test_oracle_golden.py)
need neither a GPU nor /root/reference: they pin the oracle (oracle/mc_oracle.c) to the reference's
results bit for bit; the -m gpu tests pin the HIP path to the same vectors.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.ref_lib import RefLib  # noqa: E402
from ref_pipeline import gaussian, ref_stereo_predict  # noqa: E402
from util import blocky_pair, features, raw_volumes, smooth_pair  # noqa: E402

import mc_cnn_amd as mc  # noqa: E402  (only for the PRESETS tables)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def ops_case(ref, H, W, D, C, seed):
    """One vector per hot-path operator."""
    g = {}
    x0, x1 = smooth_pair(H, W, min(D, 8), seed=seed)
    b0, b1 = blocky_pair(H, W, seed=seed + 1)
    f = features(C, H, W, seed=seed + 2)
    vl, vr = raw_volumes(D, H, W, seed=seed + 3)
    g.update(x0=x0, x1=x1, b0=b0, b1=b1, feat=f, rawL=vl, rawR=vr, dims=np.array([H, W, D, C]))
    fd = dev(f)
    jl = torch.full((1, D, H, W), float("nan"), device="cuda")
    jr = torch.full((1, D, H, W), float("nan"), device="cuda")
    ref.call("StereoJoin", fd[0:1].contiguous(), fd[1:2].contiguous(), jl, jr)
    g.update(join_L=host(jl)[0], join_R=host(jr)[0])
    for direction, tag in ((-1, "m"), (1, "p")):
        o = torch.empty((1, D, H, W), device="cuda")
        ref.call("ad", dev(x0)[None, None], dev(x1)[None, None], o, direction)
        g["ad_" + tag] = host(o)[0]
        c0, c1 = np.stack([x0, b0]), np.stack([x1, b1])
        ref.call("census", dev(c0)[None], dev(c1)[None], o, direction)
        g["census_" + tag] = host(o)[0]
    arms = {}
    for name, img, L1, tau1 in (("s", x0, 5, 0.13), ("s1", x1, 5, 0.13), ("b", b0, 14, 0.3), ("b1", b1, 14, 0.3),
                                ("z", x0, 0, 0.0)):
        o = torch.empty((1, 4, H, W), device="cuda")
        ref.call("cross", dev(img)[None], o, L1, tau1)
        arms[name] = o
        g["cross_" + name] = host(o)[0]
    g["cross_params"] = np.array([[5, 0.13], [5, 0.13], [14, 0.3], [14, 0.3], [0, 0.0]], np.float64)
    for direction, tag, vol in ((-1, "m", vl), (1, "p", vr)):
        o = torch.empty((1, D, H, W), device="cuda")
        ref.call("cbca", arms["s"], arms["s1"], dev(vol)[None], o, direction)
        g["cbca_s_" + tag] = host(o)[0]
        ref.call("cbca", arms["b"], arms["b1"], dev(vol)[None], o, direction)
        g["cbca_b_" + tag] = host(o)[0]
    sgm_prm = np.array([[4.0, 55.72, 0.02, 1.5, 3.0, 2.5], [1.3, 13.9, 0.13, 2.75, 4.5, 2.0]], np.float64)
    g["sgm_params"] = sgm_prm
    for i, prm in enumerate(sgm_prm):
        for direction, tag, vol in ((-1, "m", vl), (1, "p", vr)):
            vh = dev(vol).permute(1, 2, 0).contiguous()[None]
            o = torch.zeros((1, H, W, D), device="cuda")
            tmp = torch.empty((W, D), device="cuda")
            ref.call("sgm2", dev(x0)[None], dev(x1)[None], vh, o, tmp, *[float(v) for v in prm], direction)
            g["sgm2_%d_%s" % (i, tag)] = host(o)[0]
    am = torch.empty((1, 1, H, W), device="cuda")
    ref.call("spatial_argmin", dev(vl)[None], am)
    g["spatial_argmin_L"] = host(am)[0, 0]
    # post chain on a synthetic disparity pair
    rng = np.random.default_rng(seed + 4)
    from scipy.ndimage import gaussian_filter
    base = gaussian_filter(rng.random((H, W)), 3.0)
    base = (base - base.min()) / (np.ptp(base) + 1e-9) * (D - 1)
    d0 = np.floor(base).astype(np.float32)
    d1 = np.floor(np.roll(base, -2, axis=1)).astype(np.float32)
    noise = rng.random((H, W)) < 0.15
    d0[noise] = rng.integers(0, D, size=int(noise.sum())).astype(np.float32)
    g.update(d0=d0, d1=d1)
    outl = torch.zeros((1, 1, H, W), device="cuda")
    ref.call("outlier_detection", dev(d0)[None, None], dev(d1)[None, None], outl, D)
    occ = ref.call("interpolate_occlusion", dev(d0)[None, None], outl)[0]
    mis = ref.call("interpolate_mismatch", occ, outl)[0]
    sub = ref.call("subpixel_enchancement", mis, dev(vl)[None], D)[0]
    med = ref.call("median2d", sub, 5)[0]
    mean = ref.call("mean2d", med, gaussian(1.67).float().cuda(), 2.0)[0]
    g.update(outlier=host(outl)[0, 0], occlusion=host(occ)[0, 0], mismatch=host(mis)[0, 0], subpixel=host(sub)[0, 0],
             median5=host(med)[0, 0], mean2d=host(mean)[0, 0], mean2d_params=np.array([1.67, 2.0]))
    x = rng.standard_normal((2, C, H, W)).astype(np.float32)
    nrm = torch.empty((2, 1, H, W), device="cuda")
    o = torch.empty((2, C, H, W), device="cuda")
    ref.call("Normalize_forward", dev(x), nrm, o)
    g.update(normalize_in=x, normalize_out=host(o))
    return g


def predict_case(ref, preset, over, H, W, D, C, seed):
    prm = dict(mc.PRESETS[preset])
    prm.update(over)
    x0, x1 = smooth_pair(H, W, min(D, 10), seed=seed)
    xb = dev(np.stack([x0, x1]))[:, None]
    g = dict(x0=x0, x1=x1, dims=np.array([H, W, D, C]))
    if C:
        f = features(C, H, W, seed=seed + 1)
        g["feat"] = f
        res = ref_stereo_predict(ref, prm, xb, D, feat=dev(f))
    else:
        vl, vr = raw_volumes(D, H, W, seed=seed + 1)
        g.update(rawL=vl, rawR=vr)
        res = ref_stereo_predict(ref, prm, xb, D, raw=(dev(vl), dev(vr)))
    for k in ("volL", "volR", "dispL0", "dispR0", "disp"):
        g["out_" + k] = host(res[k]).reshape(res[k].shape[-3:] if k.startswith("vol") else res[k].shape[-2:])
    g["param_names"] = np.array(sorted(prm))
    g["param_values"] = np.array([float(prm[k]) for k in sorted(prm)], np.float64)
    return g


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    ref = RefLib()
    np.savez_compressed(os.path.join(outdir, "ops_16x40x12.npz"), **ops_case(ref, 16, 40, 12, 8, seed=100))
    np.savez_compressed(os.path.join(outdir, "ops_9x36x20.npz"), **ops_case(ref, 9, 36, 20, 3, seed=200))
    cases = [("predict_kitti_fast", "kitti_fast", {}, 20, 64, 16, 16, 300),
             ("predict_kitti_slow", "kitti_slow", {}, 20, 64, 16, 0, 400),
             ("predict_mb_slow", "mb_slow", {"cbca_i2": 4}, 20, 64, 16, 0, 500)]
    for name, preset, over, H, W, D, C, seed in cases:
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **predict_case(ref, preset, over, H, W, D, C, seed))
    for f in sorted(os.listdir(outdir)):
        print(f, os.path.getsize(os.path.join(outdir, f)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "golden"))
