#!/usr/bin/env python
"""bench.py -- throughput of the post-CNN stereo pipeline (stereo_predict, main.lua:929-1082)
on MI355X, the `-a time` analogue of the reference (main.lua:1140-1167).

A "step" is one stereo pair through the hot path (cost volume -> CBCA -> SGM -> arg-min ->
LR check -> interpolation -> sub-pixel -> median -> range-gated Gaussian) with the inputs
(normalised images + features or raw volumes) already resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config kitti_fast|kitti_slow|mb_slow]

Default workload at N = 1: BASELINE.json configs[1], KITTI 2012 fast, 370x1226, disp_max 228 (the configuration the
reference's own timing is quoted on); the 1000x1500x256 accurate configuration the north-star target is quoted on rides in
the same line as `north_star` (specified texture + two realistic pairs), KITTI accurate as `kitti_accurate`.
Default workload at N > 1: BASELINE.json configs[4] -- one Middlebury-size accurate pair per GPU, every rank its own seeded
pair (images 1234 + rank, raw volumes 7 + rank).  One rank per GPU (launched by torch.distributed.run, or spawned by this
script itself when WORLD_SIZE is not set), weak scaling, no data-path collective; the finished disparity maps are gathered
with one RCCL all-gather per step on a side stream, so that step k + 1's mc_predict overlaps step k's gather.

The timed region is EXACTLY --steps steps between barrier + device synchronisation; it is repeated (whole blocks of --steps
steps) until at least ~1 s has been timed, `ms_per_step` / `value` are the MEDIAN block, `ms_per_step_min` the best one.

Prints ONE JSON line (rank 0).  `value` = Mega-pixel-disparities / s = n_gpus * 2 volumes *
H*W*D / 1e6 / seconds-per-step.  Beside the contract's fields the line carries
  roofline      the kernel group with the largest MEASURED time among those with a byte model
                (StereoJoin / CBCA / SGM), + `kernels`: all of them
  verify        the timed configuration's outputs bit-compared with main.lua's stereo_predict over
                the REFERENCE'S OWN kernels (oracle/_ref) on the same GPU and inputs
  north_star    (default run, N=1) the 1000x1500x256 accurate configuration: ms/pair, the per-volume
                SGM + cross-aggregation sweep against SURVEY 8(d)'s 47 V budget, and its own verify
  cpu_baseline  the oracle on the host cores (min of 3)
  ops_ms_per_pair  the unchanged-main.lua route: the same pair through the op-by-op adcensus.* calls
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = {
    # name: (preset, H, W, D, C or 0 for raw volumes, BASELINE.json config string)
    "kitti_fast": ("kitti_fast", 370, 1226, 228, 64, "KITTI 2012 fast, 370x1226 disp_max=228"),
    "kitti_slow": ("kitti_slow", 370, 1226, 228, 0, "KITTI 2012 accurate (from raw volumes), 370x1226 disp_max=228"),
    "mb_slow": ("mb_slow", 1000, 1500, 256, 0, "Middlebury-size accurate (from raw volumes), 1000x1500 disp_max=256"),
    # accurate net end to end from features: FC stack (fp32 MFMA) -> fix_border -> CBCA -> SGM -> post
    "kitti_slow_fc": ("kitti_slow", 370, 1226, 228, -112, "KITTI 2012 accurate (slow net) from conv features, 370x1226 disp_max=228"),
    "tiny": ("kitti_fast", 48, 160, 32, 16, "tiny plumbing case"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
KERNEL_NAMES = {
    "join": "join_owner_kernel (StereoJoin on v_mfma_f32_32x32x2_f32, both volumes, NaN fill + fix_border folded in)",
    "cbca": "cbca_tile_kernel on real-scene arm statistics (one iteration over one volume per launch), cbca_lean2x_kernel on textures (two iterations per launch), cbca_strip_kernel for arms > 13 (the pair's route word picks on the device)",
    "sgm": "sgm_pass_kernel (right+left sweep, down sweep, up sweep: 3 launches over both volumes)",
}


def algorithmic_bytes(preset, H, W, D, C):
    """fp32, V = 4*D*H*W, F = 4*C*H*W.  Per PAIR (two volumes)."""
    V = 4.0 * D * H * W
    F = 4.0 * C * H * W
    b = dict(join=(2 * F + 2 * V) if C else 0.0,
             cbca=2 * (preset["cbca_i1"] + preset["cbca_i2"]) * 2 * V,
             sgm=2 * 11 * V * preset["sgm_i"],
             argmin=2 * V)
    b["total"] = sum(b.values())
    return b


def launches_per_step(preset, C, pair=None):
    """launches of the dominant kernel of each group per pair.  cbca: one per iteration and volume -- on the texture route
    (cbca_lean2x_kernel, 4 < L1 - 1 <= 13) one per PAIR of iterations and volume, an odd last iteration on its own."""
    i1, i2 = preset["cbca_i1"], preset["cbca_i2"]
    cb = 2 * (i1 + i2)
    if pair == "texture" and 4 < preset["L1"] - 1 <= 13:
        cb = 2 * ((i1 + 1) // 2 + (i2 + 1) // 2)
    return {"join": 1 if C else 0, "cbca": cb, "sgm": 3 * preset["sgm_i"]}


def pick_dominant(stage_ms, ab):
    """The kernel group that takes the most measured time among those SURVEY 8(d) gives a byte model for."""
    cands = [k for k in ("join", "cbca", "sgm") if ab.get(k, 0) > 0 and stage_ms.get(k, 0) > 0]
    return max(cands, key=lambda k: stage_ms[k]) if cands else None


# image pair per config: the KITTI shapes are specified on the reference's real sample pair (a committed
# fixture, tests/util.sample_pair); the 1000x1500 case is specified as a Gaussian texture (sigma 3 px), tests/util.smooth_pair,
# and gets two realistic sub-records: tests/util.natural_pair (synthetic, calibrated against the sample pair by
# tests/test_inputs.py) and the sample pair mirror-tiled to 1000x1500.  CBCA is the one stage whose cost depends on the
# pair: its additions per voxel are the support sizes (9 on a texture, ~40 on real scenes).
PAIR_OF = {"kitti_fast": "sample", "kitti_slow": "sample", "kitti_slow_fc": "sample", "mb_slow": "texture", "tiny": "texture"}
PAIR_NOTE = {"sample": "the reference's real sample pair samples/input/kittiL.png / kittiR.png (tests/golden/kitti_sample_pair.npz; mirror-tiled where the shape is not 370x1226)",
             "natural": "synthetic pair with the cross-arm statistics of the reference's real KITTI sample pair (tests/util.natural_pair)",
             "texture": "Gaussian texture, sigma 3 px, shifted by a smooth disparity field (SURVEY 8(d) recipe, tests/util.smooth_pair)",
             "mixed": "the Gaussian texture with flat patches (clipped highlights, exactly constant in both images) over 15 % of the image (tests/util.mixed_pair): the regime between the two"}


def config_key(cfg):
    return next(k for k, v in CONFIGS.items() if v is cfg or v == cfg)


def make_inputs(cfg, rank, device, pair=None, raw_planes=None):
    """This rank's inputs (device = None: host arrays only).  Seeds: images 1234 + rank (synthetic pairs), features 42 + rank,
    raw volumes 7 + rank -- BASELINE configs[4] is "8 Middlebury-size pairs, seeds 7 ... 14", one per GPU."""
    from util import features, mixed_pair, natural_pair, raw_volumes, sample_pair, smooth_pair
    preset, H, W, D, C, _ = cfg
    pair = pair or PAIR_OF[config_key(cfg)]
    if pair == "sample":   # (one real pair: every rank sees the same images; features / raw volumes are seeded per rank)
        x0, x1 = sample_pair(H, W)
    else:
        x0, x1 = {"natural": natural_pair, "mixed": mixed_pair}.get(pair, smooth_pair)(H, W, D, seed=1234 + rank)
    host = dict(x0=x0, x1=x1, seeds=dict(images=None if pair == "sample" else 1234 + rank))
    if C < 0:  # accurate net: non-negative (post-ReLU) features + a seeded FC stack (no trained nets are available)
        rng = np.random.default_rng(42 + rank)
        host["feat"] = np.maximum(rng.standard_normal((2, -C, H, W)), 0).astype(np.float32)
        host["seeds"].update(features=42 + rank, fc_stack=7 + rank)
    elif C:
        host["feat"] = features(C, H, W, seed=42 + rank)
        host["seeds"].update(features=42 + rank)
    else:
        host["raw"] = raw_volumes(raw_planes or D, H, W, seed=7 + rank)   # (raw_planes: --dry-run fingerprints the left volume's first planes only)
        host["seeds"].update(raw_volumes=7 + rank)
    if device is None:
        return None, None, host
    import torch
    xb = torch.from_numpy(np.stack([x0, x1])[:, None]).to(device)
    if C < 0:
        from mc_cnn_amd.main import load_fc
        fcl = load_fc("random:%d" % (7 + rank), "kitti")
        kw = dict(fc_feat=torch.from_numpy(host["feat"]).to(device),
                  fc_layers=[(torch.from_numpy(w).to(device), torch.from_numpy(b).to(device)) for w, b in fcl])
        host["fc_layers"] = fcl
    elif C:
        kw = dict(feat=torch.from_numpy(host["feat"]).to(device))
    else:
        kw = dict(raw=(torch.from_numpy(host["raw"][0]).to(device), torch.from_numpy(host["raw"][1]).to(device)))
    return xb, kw, host


def same_bits_dev(a, b):
    """bit-exact equality on the device with NaN == NaN (NaN masks must match)"""
    import torch
    a, b = a.reshape(-1), b.reshape(-1)
    if a.shape != b.shape:
        return False
    na, nb = torch.isnan(a), torch.isnan(b)
    bad = (na != nb) | (~na & (a.view(torch.int32) != b.view(torch.int32)))
    return not bool(bad.any().item())


def verify_against_reference(cfg, xb, kw, prm, D, ws, cfg_key):
    """Outputs of the timed call form (+ the exported volumes and arg-min maps) against main.lua's stereo_predict run
    over the reference's own kernels (oracle/_ref, test infrastructure) on this GPU, bit for bit."""
    import torch
    import mc_cnn_amd as mc
    try:
        from oracle.ref_lib import RefLib
        from ref_pipeline import ref_stereo_predict
        ref = RefLib()
    except Exception as e:  # not built (needs /root/reference at build time)
        return dict(vs="oracle/_ref", available=False, reason=str(e)[:160], bit_exact=None, config=cfg_key)
    preset, H, W, _, C, _ = cfg
    args = dict(feat=kw["feat"]) if C > 0 else dict(raw=kw["raw"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    want = ref_stereo_predict(ref, prm, xb, D, **args)
    torch.cuda.synchronize()
    ref_ms = (time.perf_counter() - t0) * 1e3
    got = mc.stereo_predict_fused(xb, prm, D, workspace=ws, want_volumes=True, want_disp0=True, **args)
    out = torch.empty((1, 1, H, W), dtype=torch.float32, device=xb.device)
    mc.stereo_predict_fused(xb, prm, D, workspace=ws, out=out, **args)   # exactly the call the timed loop makes
    torch.cuda.synchronize()
    fields = {}
    for key, label in (("volL", "left.bin"), ("volR", "right.bin"), ("dispL0", "argmin_left"), ("dispR0", "argmin_right"),
                       ("disp", "disp.bin")):
        fields[label] = same_bits_dev(got[key], want[key])
    fields["disp.bin (timed call)"] = same_bits_dev(out, want["disp"])
    return dict(vs="oracle/_ref (the reference's adcensus.cu compiled for gfx950, driven as main.lua:929-1082 does)",
                available=True, bit_exact=all(fields.values()), fields=fields, config=cfg_key,
                shape=[H, W, D], reference_ms_per_pair=round(ref_ms, 1))


def cpu_baseline(cfg, host, budget_rows, runs=3):
    """The oracle (a faithful CPU restatement of the reference, oracle/mc_oracle.c) on a bounded
    sample of the same workload: a band of `rows` image rows at full width and full disp_max; min of `runs`."""
    from oracle import cpu_oracle
    import mc_cnn_amd as mc
    preset, H, W, D, C, _ = cfg
    rows = min(H, budget_rows)
    prm = dict(mc.PRESETS[preset])
    x0, x1 = host["x0"][:rows], host["x1"][:rows]
    if C:
        kw = dict(featL=host["feat"][0][:, :rows], featR=host["feat"][1][:, :rows])
    else:
        kw = dict(rawL=host["raw"][0][:, :rows], rawR=host["raw"][1][:, :rows])
    cpu_oracle.build()
    # an untimed pass over a 4-row band first: the OpenMP pool of a 256-core host and the library's pages come up in it (the first full
    # pass of a fresh process measured 57 s against 15-24 s for the following ones); then up to `runs` timed passes inside ~40 s
    wr = min(rows, 4)
    wkw = {k: v[:, :wr] for k, v in kw.items()}
    cpu_oracle.stereo_predict(prm, x0[:wr], x1[:wr], D, **wkw)
    times = []
    for _ in range(runs):
        t0 = time.perf_counter()
        cpu_oracle.stereo_predict(prm, x0, x1, D, **kw)
        times.append(time.perf_counter() - t0)
        if sum(times) + min(times) > 40.0:
            break
    runs = len(times)
    dt = min(times)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    cores = int(os.environ.get("OMP_NUM_THREADS", cores))
    return dict(value=round(2.0 * rows * W * D / 1e6 / dt, 3), unit="MPix-disp/s", cores=cores, kind="port",
                sample="oracle stereo_predict on a %dx%dx%d band (%d of %d rows) of the same pair, min of %d runs (%s s)" %
                       (rows, W, D, rows, H, runs, "/".join("%.1f" % t for t in times)))


def stage_times(step, reps):
    acc = {}
    for _ in range(reps):
        r = step(timed=True)
        for k, v in r["stage_ms"].items():
            acc[k] = acc.get(k, 0.0) + v / reps
    acc.pop("_", None)
    return acc


_COPY_RATE = {}


def copy_rate(device, nbytes, how="torch"):
    """The box's device-to-device copy rate for a working set of the workload's size ('measure achievable
    with a device-copy kernel on the box and report against both'): read + written bytes per second over two buffers of
    nbytes each, best of 10 after 3 warm-ups.  how = "torch": dst.copy_(src), a grid-stride kernel -- what rounds 1-3 quoted
    (4.8 - 5.5 TB/s at 2 x 1.5 GB); how = "element": one 16-byte element per thread in address order (the library's mc_scale
    with factor 1), the form the guide's 6.29 TB/s is reached in (profiles/r04_bw_sizes.txt).  Boxes of the pool differ by up
    to 15 %, and buffers that fit the 256 MB Infinity Cache copy faster than HBM streams."""
    import torch
    key = (str(device), int(nbytes), how)
    if key not in _COPY_RATE:
        import mc_cnn_amd as mc
        n = max(int(nbytes) // 16 * 4, 1 << 20)
        src = torch.empty(n, dtype=torch.float32, device=device).fill_(1.0)
        dst = torch.empty_like(src)
        go = (lambda: dst.copy_(src)) if how == "torch" else (lambda: mc.adcensus.scale(src, dst, 1.0))
        for _ in range(3):
            go()
        best = float("inf")
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            go()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        _COPY_RATE[key] = 2.0 * n * 4 / (best * 1e-3) / 1e9
        del src, dst
        torch.cuda.empty_cache()
    return _COPY_RATE[key]


# scalar v_add_f32: 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T additions/s (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles);
# scripts/microbench/valu_rate.hip measures 55.8 T/s at 8 waves per SIMD and 36 T/s at 2 (profiles/r03_valu_rate.txt)
FP32_ADD_PEAK = 256 * 128 * 2.4e9


def arm_lengths(ends):
    """cross's exclusive arm ends (4,H,W) (adcensus.cu:280-322) -> lengths (left, right, up, down) as integers"""
    c = np.asarray(ends, np.float64).reshape(4, *np.shape(ends)[-2:])
    H, W = c.shape[1:]
    xs, ys = np.arange(W)[None, :], np.arange(H)[:, None]
    return np.stack([xs - c[0] - 1, c[1] - xs - 1, ys - c[2] - 1, c[3] - ys - 1]).astype(np.int64)


def support_sizes(aL, aR, d):
    """tap count of every support of the LEFT volume's plane d (adcensus.cu:343-377, direction -1: left pixel x pairs with
    right pixel x - d): per-arm minimum of the two images, rows y-up .. y+down, in each row columns x-left .. x+right.
    Returns an (H, W-d) array for the columns x >= d (the others are copied through)."""
    H, W = aL.shape[1:]
    l, r, u, dn = np.minimum(aL[:, :, d:], aR[:, :, :W - d] if d else aR)
    rowlen = l + r + 1
    cs = np.vstack([np.zeros((1, rowlen.shape[1]), np.int64), np.cumsum(rowlen, 0)])
    ys, cols = np.arange(H)[:, None], np.arange(rowlen.shape[1])[None, :]
    return cs[np.clip(ys + dn + 1, 0, H), cols] - cs[np.clip(ys - u, 0, H), cols]


def cbca_additions(xb, prm, D, n_planes=8):
    """What bounds cross-based aggregation when the supports are large: every output is ONE serial chain of additions (the
    reference's summation order), as many as its support has taps.  Mean support size of the left volume, sampled at
    n_planes disparities, from the arms the product's own `cross` computes for this pair."""
    import torch
    import mc_cnn_amd as mc
    H, W = xb.shape[-2:]
    arms = []
    for i in range(2):
        c = torch.empty((1, 4, H, W), dtype=torch.float32, device=xb.device)
        mc.adcensus.cross(xb[i:i + 1], c, prm["L1"], prm["tau1"])
        arms.append(arm_lengths(c[0].cpu().numpy()))
    taps = vox = 0
    for d in sorted({int(round(k * (D - 1) / max(1, n_planes - 1))) for k in range(n_planes)}):
        if d < W:
            size = support_sizes(arms[0], arms[1], d)
            taps += int(size.sum())
            vox += size.size
    return taps / max(1, vox)


def traffic_file(cfg_key):
    """HBM bytes per launch of a configuration's kernel groups from the rocprofv3 --pmc passes of the BUILDER's evidence run (separate passes, FETCH_SIZE /
    WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes; scripts/gpu_pmc.sh -> scripts/make_traffic_json.py): a committed constant, NOT a measurement
    of the run that prints it -- counters cannot be collected inside a timed run.  Returns (dict, what `traffic_source` says)."""
    tfile = os.path.join(ROOT, "profiles", "traffic_%s.json" % cfg_key)
    if not os.path.exists(tfile):
        return {}, None
    t = json.load(open(tfile))
    return t, "profiles/traffic_%s.json: builder's PMC pass of %s (commit %s), not measured in this run" % (cfg_key, t.get("run", "its evidence run"), t.get("commit", "n/a"))


def roofline_record(cfg_key, prm, H, W, D, C, acc, ms_per_step, device=None, xb=None, pair=None):
    ab = algorithmic_bytes(prm, H, W, D, max(C, 0))
    nl = launches_per_step(prm, max(C, 0), pair)
    traffic_all, traffic_src = traffic_file(cfg_key)
    kernels = {}
    for k in ("join", "cbca", "sgm"):
        if ab[k] > 0 and acc.get(k, 0) > 0:
            gbs = ab[k] / (acc[k] * 1e-3) / 1e9
            kernels[k] = dict(ms=round(acc[k], 4), launches=nl[k], algorithmic_GB=round(ab[k] / 1e9, 3),
                              achieved_GBs=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4))
    dom = pick_dominant(acc, ab)
    if dom is None:
        return None
    achieved = ab[dom] / (acc[dom] * 1e-3) / 1e9
    kname = KERNEL_NAMES[dom]
    if dom == "cbca":
        # the pair's route word picks on the device; what it picks for the pairs this file generates:
        if prm["L1"] <= 5:
            kname = "cbca_tile_kernel<4, ...>"
        elif pair == "texture":
            kname = ("cbca_lean2x_kernel<8, 1> (texture route: TWO iterations over one volume per launch -- the first one's rows stay in LDS --, 3 x 3 means of the "
                     "whole plane + the listed larger supports of both; cbca_classify2x once per pair and direction)")
        elif pair is not None:
            kname = "cbca_tile_kernel<13, ...> (real-scene arm statistics)"
        else:
            kname = "cbca_tile_kernel<13, ...> (real-scene arm statistics) or cbca_lean2x_kernel (texture), by the pair's route word"
        if pair != "texture" or prm["L1"] <= 5:
            kname += " (one iteration over one volume per launch)"
    rec = dict(bound="hbm", kernel=kname, picked_by="largest measured stage time", achieved=round(achieved, 1),
               peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
               # (profiles/traffic_*.json holds cbca's bytes per ITERATION; per launch like `achieved`: x the iterations a launch runs)
               traffic=(round(traffic_all[dom] * (2 * (prm["cbca_i1"] + prm["cbca_i2"]) / nl[dom])) if dom == "cbca" and traffic_all.get(dom) else traffic_all.get(dom)),
               traffic_source=traffic_src if traffic_all.get(dom) else None,
               launches_per_step=nl[dom], algorithmic_bytes_per_launch=round(ab[dom] / nl[dom]),
               avg_launch_ms=round(acc[dom] / nl[dom], 4), kernels=kernels,
               pipeline_frac=round(ab["total"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    if xb is not None and "cbca" in kernels:
        try:  # the compute side of cbca: serial additions per voxel on THIS pair
            apv = cbca_additions(xb, prm, D)
            adds = apv * 2.0 * D * H * W * (prm["cbca_i1"] + prm["cbca_i2"])   # both volumes, all iterations (NaN-triangle voxels are copies: slight overcount)
            t_hbm, t_add = ab["cbca"] / (HBM_PEAK_GBS * 1e9), adds / FP32_ADD_PEAK   # seconds at either peak
            kernels["cbca"].update(additions_per_voxel=round(apv, 2), additions_T_per_s=round(adds / (acc["cbca"] * 1e-3) / 1e12, 3),
                                   frac_of_fp32_add_peak=round(adds / (acc["cbca"] * 1e-3) / FP32_ADD_PEAK, 4),
                                   bound_ms=dict(hbm=round(t_hbm * 1e3, 3), fp32_add=round(t_add * 1e3, 3)),
                                   frac_two_bounds=round(max(t_hbm, t_add) / (acc["cbca"] * 1e-3), 4),
                                   additions_note="mean support size of the left volume at 8 disparities; peak = 78.6 T scalar fp32 additions/s (55.8 T/s measured on a pure chain of v_add_f32)")
        except Exception as e:  # never lose the bench line over a side figure
            kernels["cbca"]["additions_error"] = str(e)[:200]
    if dom == "cbca" and "frac_two_bounds" in kernels["cbca"]:
        # cross-based aggregation has two rooflines: 2 V of bytes per iteration and one serial fp32 addition per support tap;
        # frac (above) is against HBM as the contract asks, this one against whichever bound is the tighter on this pair
        rec["frac_two_bounds"] = kernels["cbca"]["frac_two_bounds"]
        rec["bound_ms"] = kernels["cbca"]["bound_ms"]
    if device is not None:  # the same figures against what a plain copy of one volume reaches on THIS box
        cr, ce = copy_rate(device, 4 * D * H * W), copy_rate(device, 4 * D * H * W, "element")
        rec["box_copy"] = dict(GBs=round(ce, 1), torch_copy_GBs=round(cr, 1), working_set_bytes=2 * 4 * D * H * W, frac_of_peak=round(ce / HBM_PEAK_GBS, 4),
                               achieved_over_copy=round(achieved / ce, 4),
                               note="two buffers of one volume each, best of 10: GBs = one 16-byte element per thread in address order (mc_scale), "
                                    "torch_copy_GBs = dst.copy_(src), a grid-stride kernel (the figure rounds 1-3 quoted)")
    return rec


def sub_record(device, config, reps=10, pair=None):
    """A second single-GPU configuration inside the default line (driver-timed instead of builder-kept): ms per pair, stage times,
    its dominant kernel group against the HBM roofline, verify against the reference's kernels."""
    import torch
    import mc_cnn_amd as mc
    from mc_cnn_amd.predict import Workspace
    cfg = CONFIGS[config]
    preset, H, W, D, C, name = cfg
    prm = dict(mc.PRESETS[preset])
    pair = pair or PAIR_OF[config]
    xb, kw, _ = make_inputs(cfg, 0, device, pair)
    ws = Workspace(prm, D, H, W, device)
    out = torch.empty((1, 1, H, W), dtype=torch.float32, device=device)

    def step(timed=False):
        return mc.stereo_predict_fused(xb, prm, D, workspace=ws, out=out, timed=timed, **kw)
    for _ in range(3):
        step()
    times = []
    for _ in range(reps):   # (min of N, as main.lua:1152-1167 reports its timing)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    acc = stage_times(step, 3)
    n_it = prm["cbca_i1"] + prm["cbca_i2"]
    rec = dict(workload=name, pair=PAIR_NOTE[pair], ms_per_pair=round(float(np.median(times)), 4), ms_per_pair_min=round(min(times), 4),
               value_MPix_disp_s=round(2.0 * H * W * D / 1e6 / (float(np.median(times)) * 1e-3), 1),
               stage_ms={k: round(v, 4) for k, v in acc.items()},
               cbca_ms_per_launch=round(acc.get("cbca", 0) / max(1, launches_per_step(prm, max(C, 0), pair)["cbca"]), 4),
               cbca_ms_per_iteration=round(acc.get("cbca", 0) / max(1, 2 * n_it), 4),
               roofline=roofline_record(config, prm, H, W, D, C, acc, float(np.median(times)), None, xb, pair),
               verify=verify_against_reference(cfg, xb, kw, prm, D, ws, config))
    del ws, xb, kw
    torch.cuda.empty_cache()
    return rec


def conv_pmc():
    """the convolution's matrix-pipe counters from the builder's PMC pass (they cannot be collected inside a timed run), labelled as such"""
    path = os.path.join(ROOT, "profiles", "conv3x3_mfma_busy.json")
    try:
        j = json.load(open(path))
        k = j["kitti_64"]
        return dict(mfma_busy_frac_of_launch=k["mfma_busy_frac_of_launch"], clock_GHz_while_profiled=k["clock_GHz_while_profiled"],
                    mfma_busy_source="profiles/conv3x3_mfma_busy.json: builder's PMC pass of %s (commit %s), not measured in this run" % (j["run"], j["commit"]))
    except Exception:
        return dict(mfma_busy_frac_of_launch=None)


def from_images_record(device, reps=20):
    """KITTI-2012 fast from the IMAGES (main.lua:1084-1100 starts at the PNGs): normalised (2,1,370,1226) pair -> feature net
    (4 x mc_conv3x3 + mc_normalize_forward, seeded weights resident on the device) -> mc_predict.  Per-stage HIP-event times on the launch
    stream, the 64->64 convolutions against the fp32-MFMA peak, features checked against a float64 convolution (tolerance 1e-4: cuDNN's
    order is unpinned) and the disparity map bit-compared with the reference's kernels fed the SAME features."""
    import torch
    import mc_cnn_amd as mc
    from mc_cnn_amd import adcensus
    from mc_cnn_amd import main as mcmain
    from mc_cnn_amd.predict import Workspace
    cfg = CONFIGS["kitti_fast"]
    preset, H, W, D, C, name = cfg
    prm = dict(mc.PRESETS[preset])
    xb, _, _ = make_inputs(cfg, 0, device, "sample")
    layers = mcmain.device_layers(mcmain.load_net("random:42", "kitti", "fast"), device)
    prm["border_n"] = len(layers)
    ws = Workspace(prm, D, H, W, device)
    out = torch.empty((1, 1, H, W), dtype=torch.float32, device=device)

    def step():
        return mc.stereo_predict_fused(xb, prm, D, feat=mcmain.features_fast(xb, layers), workspace=ws, out=out)
    for _ in range(3):
        step()
    times = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    # stages: events on the stream the kernels are launched on (torch's current stream)
    stage = {}
    for _ in range(5):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(len(layers) + 3)]
        h = xb
        marks[0].record()
        for i, (w, b) in enumerate(layers):
            h = adcensus.conv3x3(h, w, b, relu=i < len(layers) - 1)
            marks[i + 1].record()
        norm = torch.empty((2, 1, H, W), dtype=torch.float32, device=device)
        feat = torch.empty_like(h)
        adcensus.Normalize_forward(h, norm, feat)
        marks[len(layers) + 1].record()
        mc.stereo_predict_fused(xb, prm, D, feat=feat, workspace=ws, out=out)
        marks[len(layers) + 2].record()
        torch.cuda.synchronize()
        names = ["conv%d" % (i + 1) for i in range(len(layers))] + ["normalize", "predict"]
        for k, nm in enumerate(names):
            stage[nm] = stage.get(nm, 0.0) + marks[k].elapsed_time(marks[k + 1]) / 5
    fm = layers[1][0].shape[0]
    inner = [stage["conv%d" % (i + 1)] for i in range(1, len(layers))]
    flop = 2.0 * 2 * H * W * fm * fm * 9
    tf = flop / (float(np.mean(inner)) * 1e-3) / 1e12
    # parity of the features: float64 convolution chain (torch) on the same weights
    import torch.nn.functional as F
    h64 = xb.double()
    for i, (w, b) in enumerate(layers):
        h64 = F.conv2d(h64, w.double(), b.double(), padding=1)
        if i < len(layers) - 1:
            h64 = F.relu(h64)
    want = h64 / torch.sqrt((h64 * h64).sum(1, keepdim=True) + 1e-5)
    feat = mcmain.features_fast(xb, layers)
    err = float((feat.double() - want).abs().max())
    ver = verify_against_reference(cfg, xb, dict(feat=feat), prm, D, ws, "kitti_fast_from_images")
    med = float(np.median(times))
    rec = dict(workload="KITTI 2012 fast from the normalised image pair: 4 x conv3x3 (1->64, 3 x 64->64) + Normalize2 + mc_predict, 370x1226 disp_max=228, seeded weights",
               ms_per_pair=round(med, 4), ms_per_pair_min=round(min(times), 4),
               value_MPix_disp_s=round(2.0 * H * W * D / 1e6 / (med * 1e-3), 1),
               stage_ms={k: round(v, 4) for k, v in stage.items()},
               roofline=dict(bound="mfma", kernel="conv3x3_kernel<2> (64->64, both images per launch; resident filter bank, v_mfma_f32_32x32x2_f32)",
                             achieved=round(tf, 1), peak=157.3, unit="TFLOP/s", frac=round(tf / 157.3, 4), traffic=None,
                             algorithmic_flops_per_launch=flop, avg_launch_ms=round(float(np.mean(inner)), 4),
                             note="flops = 2*N*H*W*Cin*Cout*9; launch time includes the 5 us re-layout of the weights", **conv_pmc()),
               features_max_abs_err_vs_float64=err, features_tolerance=1e-4, features_ok=bool(err <= 1e-4),
               verify=dict(bit_exact=ver.get("bit_exact"), available=ver.get("available"),
                           note="mc_predict on these features against the reference's kernels on the same features"))
    del ws, xb
    torch.cuda.empty_cache()
    return rec


def pipelined_record(device, config, ks=(1, 2, 3), steps=24, from_images=False):
    """K pairs in flight: K workspaces, K streams, mc_predict (and the feature net with from_images) issued round-robin.  The stages of a
    pair are bound by different units (SGM: memory; StereoJoin / convolutions: matrix pipe; post-processing, tile kernel: issue), one pair
    on one stream runs them strictly one after another.  Each slot has its own seeded inputs; every slot's map is bit-compared with the same
    inputs run alone."""
    import torch
    import mc_cnn_amd as mc
    from mc_cnn_amd import main as mcmain
    from mc_cnn_amd.predict import Workspace
    cfg = CONFIGS[config]
    preset, H, W, D, C, name = cfg
    prm = dict(mc.PRESETS[preset])
    kmax = max(ks)
    layers = None
    if from_images:
        layers = mcmain.device_layers(mcmain.load_net("random:42", "kitti", "fast"), device)
        prm["border_n"] = len(layers)
    slots = []
    for r in range(kmax):
        xb, kw, _ = make_inputs(cfg, r, device, PAIR_OF[config])
        slots.append(dict(xb=xb, kw=kw, ws=Workspace(prm, D, H, W, device), out=torch.empty((1, 1, H, W), dtype=torch.float32, device=device),
                          stream=torch.cuda.Stream(device=device)))

    def one(sl):
        if from_images:
            return mc.stereo_predict_fused(sl["xb"], prm, D, feat=mcmain.features_fast(sl["xb"], layers), workspace=sl["ws"], out=sl["out"])
        return mc.stereo_predict_fused(sl["xb"], prm, D, workspace=sl["ws"], out=sl["out"], **sl["kw"])
    alone = []
    for sl in slots:   # every slot alone, on the default stream: the maps the pipelined runs must reproduce
        one(sl)
        torch.cuda.synchronize()
        alone.append(sl["out"].clone())
    res = {}
    for K in ks:
        def block(n):
            for i in range(n):
                sl = slots[i % K]
                with torch.cuda.stream(sl["stream"]):
                    one(sl)
        block(2 * K)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            block(steps)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / steps * 1e3)
        ok = all(same_bits_dev(slots[k]["out"], alone[k]) for k in range(K))
        ms = float(np.median(ts))
        res["K%d" % K] = dict(ms_per_pair=round(ms, 4), ms_per_pair_min=round(min(ts), 4), value_MPix_disp_s=round(2.0 * H * W * D / 1e6 / (ms * 1e-3), 1),
                              bit_exact_vs_alone=bool(ok))
    base = res["K%d" % min(ks)]["ms_per_pair"]
    best = min(res, key=lambda k: res[k]["ms_per_pair"])
    rec = dict(workload=name + (" from the image pair (feature net included)" if from_images else ""), steps_per_block=steps,
               pairs_in_flight=res, best=best, pipeline_frac=round(res[best]["ms_per_pair"] / base, 4),
               note="K streams x K workspaces, round-robin; ms_per_pair = wall time of a block / pairs in it, median of 5 blocks; pipeline_frac = best / K1")
    del slots
    torch.cuda.empty_cache()
    return rec


def fc_stack_record(device):
    """The accurate architecture's cost-volume stage (mc_fc_stack, fp32 MFMA) at 370x1226x228, both volumes from one pass: ONE timed
    call after one warm-up (0.76 s each), so that the driver's default run times it once; `--config kitti_slow_fc` is the full line."""
    import torch
    import mc_cnn_amd as mc
    from mc_cnn_amd.fc import fc_cost_volumes
    cfg = CONFIGS["kitti_slow_fc"]
    preset, H, W, D, C, name = cfg
    xb, kw, _ = make_inputs(cfg, 0, device)
    need = mc._lib.lib.mc_fc_stack_workspace_bytes(-C, len(kw["fc_layers"]), H, W)
    fc_ws = torch.empty(need + 16, dtype=torch.uint8, device=device)
    times = []
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        vl, vr = fc_cost_volumes(kw["fc_feat"], kw["fc_layers"], D, workspace=fc_ws)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    dims = [w.shape[1] for w, _ in kw["fc_layers"]] + [1]
    vox = sum(max(0, W - d) for d in range(D)) * H
    flop = vox * 2.0 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))  # the reference's flops (main.lua:958-983)
    ms = times[-1]
    tf = flop / (ms * 1e-3) / 1e12
    finite = bool(torch.isfinite(vl[0, 0, :, D:]).all().item())   # (a sanity bit, not parity: tests/test_gpu_fc.py holds the 1e-4 comparison)
    rec = dict(workload=name, ms_per_call=round(ms, 2), warmup_ms=round(times[0], 2), timed_calls=1,
               roofline=dict(bound="mfma", kernel="fc_stack_kernel (+ fc_project_kernel): both volumes from one pass", achieved=round(tf, 1), peak=157.3,
                             unit="TFLOP/s", frac=round(tf / 157.3, 4), algorithmic_flops_per_launch=flop,
                             note="fp32 MFMA (v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s dense peak); flops = the reference's count, the executed ones are 16 % fewer"),
               outputs_finite=finite, parity="tolerance 1e-4 against the oracle and an fp32 addmm chain: tests/test_gpu_fc.py (BLAS summation order unpinned)")
    del xb, kw, fc_ws, vl, vr
    torch.cuda.empty_cache()
    return rec


def north_star_record(device, steps=5, with_cpu=True):
    """BASELINE.json north_star: the SGM + cross-aggregation sweep at 1500x1000x256 (mb-slow parameters,
    main.lua:132-144: 2 + 16 CBCA iterations), per volume, against SURVEY 8(d)'s 47 V = 72.2 GB budget."""
    import torch
    import mc_cnn_amd as mc
    from mc_cnn_amd.predict import Workspace
    cfg = CONFIGS["mb_slow"]
    preset, H, W, D, C, name = cfg
    prm = dict(mc.PRESETS[preset])
    xb, kw, host = make_inputs(cfg, 0, device)
    ws = Workspace(prm, D, H, W, device)
    out = torch.empty((1, 1, H, W), dtype=torch.float32, device=device)

    def step(timed=False):
        return mc.stereo_predict_fused(xb, prm, D, workspace=ws, out=out, timed=timed, **kw)
    for _ in range(2):  # the first pass over a fresh 9 GB workspace pays its page mapping
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    acc = stage_times(step, 3)
    V = 4.0 * D * H * W
    n_it = prm["cbca_i1"] + prm["cbca_i2"]
    sweep_ms = (acc.get("cbca", 0) + acc.get("sgm", 0)) / 2          # per volume
    layout_ms = acc.get("layout", 0) / 2
    budget = (2 * n_it + 11) * V
    rec = dict(workload=name, H=H, W=W, disp_max=D, cbca_iterations=n_it, ms_per_pair=round(ms, 3),
               value_MPix_disp_s=round(2.0 * H * W * D / 1e6 / (ms * 1e-3), 1),
               stage_ms={k: round(v, 3) for k, v in acc.items()},
               per_volume=dict(cbca_ms=round(acc.get("cbca", 0) / 2, 3), sgm_ms=round(acc.get("sgm", 0) / 2, 3),
                               layout_ms=round(layout_ms, 3), algorithmic_GB=round(budget / 1e9, 2),
                               sweep_ms=round(sweep_ms, 3),
                               frac_of_hbm_peak=round(budget / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               frac_incl_layout=round(budget / ((sweep_ms + layout_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               target=0.70),
               roofline=roofline_record("mb_slow", prm, H, W, D, C, acc, ms, device, xb, "texture"),
               verify=verify_against_reference(cfg, xb, kw, prm, D, ws, "mb_slow"))
    del ws, xb, kw
    torch.cuda.empty_cache()
    cr = copy_rate(device, V, "element")   # what a plain copy of one volume reaches on this box (one 16-byte element per thread, address order)
    rec["per_volume"]["box_copy_GBs"] = round(cr, 1)
    rec["per_volume"]["box_torch_copy_GBs"] = round(copy_rate(device, V), 1)
    rec["per_volume"]["sweep_over_box_copy"] = round(budget / (sweep_ms * 1e-3) / 1e9 / cr, 4)
    rec["pair"] = PAIR_NOTE["texture"]
    if with_cpu:   # the oracle on a 64-row band of this very workload (full width, full disp_max, 2 + 16 iterations): ~20-30 s of host time
        rec["cpu_baseline"] = cpu_baseline(cfg, host, 64, runs=1)
    rec["realistic_pair"] = north_star_realistic(device, "natural")
    rec["realistic_pair_sample"] = north_star_realistic(device, "sample")
    rec["mixed_pair"] = north_star_realistic(device, "mixed")
    return rec


def north_star_realistic(device, pair):
    """The same sweep on a pair with real-scene arm statistics: cross-based aggregation does ~45 additions per voxel there
    (half of the supports minimal, 4-6 % flat regions of up to 27 x 27 taps) instead of 9 -- it is bound by the serial
    additions the reference's summation order imposes, not by HBM.  One pair, reference check included."""
    import torch
    import mc_cnn_amd as mc
    from mc_cnn_amd.predict import Workspace
    cfg = CONFIGS["mb_slow"]
    preset, H, W, D, C, name = cfg
    prm = dict(mc.PRESETS[preset])
    xb, kw, _ = make_inputs(cfg, 0, device, pair=pair)
    ws = Workspace(prm, D, H, W, device)
    out = torch.empty((1, 1, H, W), dtype=torch.float32, device=device)

    def step(timed=False):
        return mc.stereo_predict_fused(xb, prm, D, workspace=ws, out=out, timed=timed, **kw)
    step()
    acc = stage_times(step, 1)
    n_it = prm["cbca_i1"] + prm["cbca_i2"]
    try:
        apv = cbca_additions(xb, prm, D)
    except Exception:  # a side figure must not cost the record
        apv = float("nan")
    adds = apv * 2.0 * D * H * W * n_it
    ops_ms = None
    try:   # the unchanged-main.lua route (op-by-op adcensus.* calls) on the same pair: adcensus.cbca picks the same kernel on the device
        mc.stereo_predict(xb, prm, D, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mc.stereo_predict(xb, prm, D, **kw)
        torch.cuda.synchronize()
        ops_ms = round((time.perf_counter() - t0) * 1e3, 2)
    except Exception as e:
        ops_ms = "error: %s" % str(e)[:200]
    tall, tsrc = traffic_file("mb_slow_%s" % pair)
    traffic = tall.get("cbca")
    rec = dict(pair=PAIR_NOTE[pair], ms_per_pair=round(sum(acc.values()), 2), stage_ms={k: round(v, 3) for k, v in acc.items()},
               ops_ms_per_pair=ops_ms, cbca_traffic=traffic, cbca_traffic_source=tsrc if traffic else None,
               cbca_ms_per_launch=round(acc.get("cbca", 0) / (2 * n_it), 3), cbca_additions_per_voxel=round(apv, 2),
               cbca_additions_T_per_s=round(adds / (acc["cbca"] * 1e-3) / 1e12, 3),
               cbca_frac_of_fp32_add_peak=round(adds / (acc["cbca"] * 1e-3) / FP32_ADD_PEAK, 4),
               verify=verify_against_reference(cfg, xb, kw, prm, D, ws, "mb_slow"))
    del ws, xb, kw
    torch.cuda.empty_cache()
    return rec


def rank_cpus(rank, world):
    """The host cores of rank `rank`: an equal, contiguous share of the cores this process may run on (consecutive core ids
    share a NUMA node on the GPU boxes, so the share of rank r is local to GPU r's half of the machine when the launcher
    enumerates devices in bus order).  One mc_predict step is ~40 launches issued by ONE thread: what matters is that the
    ranks' launch threads do not migrate or share a core."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if world <= 1 or len(cpus) < world:
        return None
    per = len(cpus) // world
    return cpus[rank * per:(rank + 1) * per]


def pin_rank(rank, world):
    cpus = rank_cpus(rank, world)
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            return None
    return cpus


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, p.wait())
    return rc


def pick_device(local_rank, one_gpu):
    """cuda device index of this rank.  Launchers that narrow the visible devices per rank (HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES = one device) leave exactly one device, index 0; otherwise LOCAL_RANK."""
    import torch
    n = torch.cuda.device_count()
    if one_gpu or n == 1:
        return 0
    if local_rank >= n:
        raise SystemExit("bench.py: LOCAL_RANK %d but only %d visible devices" % (local_rank, n))
    return local_rank


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: kitti_fast (BASELINE configs[1]) on one GPU, mb_slow per rank (configs[4]) on several")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the CPU-baseline band (0 = auto)")
    ap.add_argument("--no-ref-gpu", action="store_true", help="skip the verify leg (reference's own kernels on this GPU)")
    ap.add_argument("--no-north-star", action="store_true", help="skip the 1000x1500x256 and KITTI-accurate sub-records of the default run")
    ap.add_argument("--no-ops", action="store_true", help="skip timing the op-by-op (unchanged main.lua) route")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="repeat the timed block of --steps steps until this much has been timed")
    ap.add_argument("--dry-run", action="store_true", help="build this rank's inputs, print their fingerprint as JSON and exit (no GPU needed)")
    ap.add_argument("--pairs-in-flight", type=int, default=0,
                    help="K > 0: only the `pipelined` record of --config with 1..K pairs in flight (K streams, K workspaces), as one JSON line")
    ap.add_argument("--pair", choices=("sample", "natural", "texture", "mixed"), default=None,
                    help="image pair (default per config, PAIR_OF): real-scene arm statistics or the Gaussian texture")
    args = ap.parse_args()
    if args.config is None:
        args.config = "kitti_fast" if args.gpus <= 1 else "mb_slow"

    if args.dry_run:   # what rank r would process: the C5 inputs must differ per rank (tests/test_bench_contract.py)
        import hashlib
        rank = int(os.environ.get("RANK", "0"))
        cfg = CONFIGS[args.config]
        _, _, host = make_inputs(cfg, rank, None, args.pair or PAIR_OF[args.config], raw_planes=2)
        fp = {k: hashlib.sha1(np.ascontiguousarray(v if not isinstance(v, tuple) else v[0][:2]).tobytes()).hexdigest()[:16]
              for k, v in host.items() if k in ("x0", "x1", "raw", "feat")}   # (raw volumes: the first two planes of the left one)
        print(json.dumps(dict(config=args.config, workload=cfg[5], rank=rank, gpus=args.gpus, seeds=host["seeds"], inputs=fp,
                              cpus=rank_cpus(rank, args.gpus))))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    pinned = pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world))) if world > 1 else None   # before torch starts its threads

    import torch
    import torch.distributed as dist
    import mc_cnn_amd as mc
    from mc_cnn_amd.predict import Workspace

    # MC_BENCH_ONE_GPU=1 (plumbing check on a 1-GPU box only): every rank uses cuda:0 and the collectives run over gloo
    one_gpu = os.environ.get("MC_BENCH_ONE_GPU") == "1"
    dev_index = pick_device(local_rank, one_gpu)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    if args.pairs_in_flight > 0 and world == 1:
        print(json.dumps(dict(pipelined=pipelined_record(device, args.config, ks=tuple(range(1, args.pairs_in_flight + 1))))))
        return

    cfg = CONFIGS[args.config]
    preset_name, H, W, D, C, cfg_name = cfg
    prm = dict(mc.PRESETS[preset_name])
    pair = args.pair or PAIR_OF[args.config]
    xb, kw, host = make_inputs(cfg, rank, device, pair)
    ws = Workspace(prm, D, H, W, device)
    # two result maps: while the side stream gathers step k's map, step k + 1 writes the other one
    outs = [torch.empty((1, 1, H, W), dtype=torch.float32, device=device) for _ in range(2 if world > 1 else 1)]
    out = outs[0]
    gathered = torch.empty((world, H, W), dtype=torch.float32, device=device) if world > 1 else None
    side = torch.cuda.Stream(device=device) if world > 1 else None
    gather_done = [None, None]   # per result map: the event after which it may be overwritten

    fc_ws = None
    if "fc_feat" in kw:
        from mc_cnn_amd.fc import fc_cost_volumes
        need = mc._lib.lib.mc_fc_stack_workspace_bytes(-C, len(kw["fc_layers"]), H, W)
        fc_ws = torch.empty(need + 16, dtype=torch.uint8, device=device)

    def cost_volume():
        """the accurate net's cost-volume stage: FC stack + fix_border (main.lua:958-983)"""
        vl, vr = fc_cost_volumes(kw["fc_feat"], kw["fc_layers"], D, workspace=fc_ws)
        mc.adcensus.fix_border(vl, prm["border_n"], -1)
        mc.adcensus.fix_border(vr, prm["border_n"], 1)
        return vl, vr

    def step(timed=False, o=None):
        o = out if o is None else o
        if fc_ws is not None:
            return mc.stereo_predict_fused(xb, prm, D, workspace=ws, out=o, raw=cost_volume(), timed=timed)
        return mc.stereo_predict_fused(xb, prm, D, workspace=ws, out=o, timed=timed, **kw)

    def gather(o=None):
        # the path's only exchange: finished disparity maps (H*W*4 B per GPU) over xGMI
        o = out if o is None else o
        if one_gpu:
            parts = [torch.empty((1, H, W)) for _ in range(world)]
            dist.all_gather(parts, o.view(1, H, W).cpu())
            gathered.copy_(torch.cat(parts).to(device))
        else:
            dist.all_gather_into_tensor(gathered, o.view(1, H, W))

    def step_and_gather(k):
        """step k: mc_predict on the compute stream into map k % 2, its all-gather on the side stream"""
        if world == 1:
            step()
            return
        o = outs[k & 1]
        if gather_done[k & 1] is not None:
            torch.cuda.current_stream().wait_event(gather_done[k & 1])   # the gather of step k - 2 has read this map
        step(o=o)
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            gather(o)
            ev = torch.cuda.Event()
            ev.record(side)
        gather_done[k & 1] = ev

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step_and_gather(k)
    # the timed region: EXACTLY --steps steps between barrier + synchronise on both sides, max over ranks; repeated as whole
    # blocks until >= --min-seconds have been timed (20 KITTI-fast steps are 58 ms: clocks, first touches and box-to-box spread
    # would all sit inside one blink), median block reported, best block beside it
    blocks = []
    while True:
        sync()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step_and_gather(k)
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        blocks.append(dt)
        if sum(blocks) >= args.min_seconds or len(blocks) >= 200:
            break
    dt = float(np.median(blocks))
    ms_per_step = dt / args.steps * 1e3
    value = world * 2.0 * H * W * D / 1e6 / (dt / args.steps)

    # ---- multi-rank bookkeeping: who took part, and is the gathered batch what the ranks computed ----
    multi = None
    if world > 1:
        ids = torch.tensor([rank], dtype=torch.int64, device="cpu" if one_gpu else device)
        seen = [torch.zeros_like(ids) for _ in range(world)]
        dist.all_gather(seen, ids)
        ranks_seen = sorted(int(s.item()) for s in seen)
        sync()   # (the side stream's last gathers are done)
        step()
        gather()
        torch.cuda.synchronize()
        own_ok = same_bits_dev(gathered[rank], out)            # every rank: its slot holds its own result
        flag = torch.tensor([1 if own_ok else 0], dtype=torch.int64, device="cpu" if one_gpu else device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        checked, others_ok = [], True
        if rank == 0 and fc_ws is None:
            # rank 0 recomputes the pairs of up to two other ranks on its own GPU: the gathered maps must equal
            # what a single-GPU run produces for those pairs
            for r in sorted({1, world - 1}):
                xb_r, kw_r, _ = make_inputs(cfg, r, device)
                o = mc.stereo_predict_fused(xb_r, prm, D, workspace=ws, **kw_r)["disp"]
                torch.cuda.synchronize()
                others_ok = others_ok and same_bits_dev(gathered[r], o)
                checked.append(r)
                del xb_r, kw_r
        # where a step's time goes on every rank (outside the timed region): compute and gather timed separately, each behind
        # a barrier so that a rank's figure is its own work and not its wait for the slowest peer; and the spread of the
        # ranks' compute times (launch skew / a slow GPU shows up here, not in the aggregate)
        nrep = max(3, min(10, args.steps))
        comp = gath = 0.0
        for _ in range(nrep):
            sync()
            t1 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            dist.barrier()
            t3 = time.perf_counter()
            gather()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            comp += (t2 - t1) * 1e3 / nrep
            gath += (t4 - t3) * 1e3 / nrep
        mine = torch.tensor([comp, gath], dtype=torch.float64, device="cpu" if one_gpu else device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        comp_all = [round(float(t[0]), 4) for t in allr]
        gath_all = [round(float(t[1]), 4) for t in allr]
        multi = dict(ranks_seen=ranks_seen, world_size=world, backend="gloo (MC_BENCH_ONE_GPU)" if one_gpu else "nccl (RCCL)",
                     own_slot_bit_exact_all_ranks=bool(flag.item()), ranks_recomputed_on_rank0=checked,
                     gathered_equals_single_gpu=bool(others_ok),
                     per_rank_compute_ms=comp_all, per_rank_gather_ms=gath_all,
                     compute_skew_ms=round(max(comp_all) - min(comp_all), 4),
                     gather_overlapped=True, rank_cpus="%d cores per rank, contiguous shares" % len(pinned) if pinned else None,
                     inputs="every rank its own pair: images seed 1234 + rank, raw volumes / features seed 7 + rank / 42 + rank" if pair != "sample"
                            else "one real image pair on every rank, raw volumes / features seeded per rank",
                     note="per-rank means over %d untimed steps: compute = one pair through mc_predict (device-synchronised), gather = "
                          "the all-gather of the (H,W) maps entered by all ranks together (here on the compute stream; in the timed "
                          "loop it runs on a side stream under the next step's mc_predict)" % nrep)

    # live per-stage HIP-event timing (same stream) for the roofline
    roof = stage = None
    acc = {}
    if rank == 0:
        reps = max(3, min(10, args.steps))
        if fc_ws is not None:  # the FC stack runs before mc_predict: time it with events on the same stream
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                cost_volume()
                e1.record()
                torch.cuda.synchronize()
                acc["fc_stack"] = acc.get("fc_stack", 0.0) + e0.elapsed_time(e1) / reps
        acc.update(stage_times(step, reps))
        stage = {k: round(v, 4) for k, v in acc.items()}
        roof = roofline_record(args.config + ("" if pair == PAIR_OF[args.config] else "_" + pair), prm, H, W, D, C, acc, ms_per_step, device, xb, pair)

    if rank == 0 and fc_ws is not None and roof is not None:
        # the accurate net's dominant kernel is the FC stack: a dense fp32 GEMM chain on the matrix cores
        dims = [w.shape[1] for w, _ in kw["fc_layers"]] + [1]
        vox = sum(max(0, W - d) for d in range(D)) * H
        flop = vox * 2.0 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))  # the reference's flops (main.lua:958-983)
        tf = flop / (acc["fc_stack"] * 1e-3) / 1e12
        roof = dict(bound="mfma", kernel="fc_stack_kernel (+ fc_project_kernel): both volumes from one pass", achieved=round(tf, 1),
                    peak=157.3, unit="TFLOP/s", frac=round(tf / 157.3, 4), traffic=None, launches_per_step=1,
                    algorithmic_flops_per_launch=flop, avg_launch_ms=round(acc["fc_stack"], 3), kernels=roof.get("kernels"),
                    note="fp32 MFMA (v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s dense peak); layer 1 is evaluated as two per-pixel "
                         "projections, so the executed flops are 16 % below the reference's count used here")

    verify = ops_ms = None
    if rank == 0 and world == 1 and fc_ws is None:
        if not args.no_ref_gpu:
            verify = verify_against_reference(cfg, xb, kw, prm, D, ws, args.config)
        if not args.no_ops:
            # the route an unchanged main.lua takes through the shim: ~25 adcensus.* calls + tensor glue per pair
            mc.stereo_predict(xb, prm, D, **kw)
            torch.cuda.synchronize()
            n = 3
            t0 = time.perf_counter()
            for _ in range(n):
                mc.stereo_predict(xb, prm, D, **kw)
            torch.cuda.synchronize()
            ops_ms = round((time.perf_counter() - t0) / n * 1e3, 3)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config != "kitti_slow_fc":
        # (kitti_slow_fc: the oracle's FC stack is a scalar triple loop, hours at this size; see kitti_slow)
        rows = args.cpu_rows or {"kitti_fast": 370, "kitti_slow": 370, "mb_slow": 64, "tiny": 48}[args.config]
        cpu = cpu_baseline(cfg, host, rows)

    north = kacc = kfc = fimg = pipe = None
    if rank == 0 and world == 1 and args.config == "kitti_fast" and not args.no_north_star:
        del ws, xb, kw
        torch.cuda.empty_cache()
        for name, fn in (("fimg", lambda: from_images_record(device)),
                         ("pipe", lambda: dict(kitti_fast=pipelined_record(device, "kitti_fast"),
                                               kitti_fast_from_images=pipelined_record(device, "kitti_fast", ks=(1, 2), from_images=True),
                                               kitti_accurate=pipelined_record(device, "kitti_slow", ks=(1, 2), steps=12)))):
            try:
                val = fn()
            except Exception as e:   # (a side record must not cost the line)
                val = dict(error=str(e)[:300])
            if name == "fimg":
                fimg = val
            else:
                pipe = val
        kacc = sub_record(device, "kitti_slow", reps=10)
        north = north_star_record(device, with_cpu=not args.no_cpu_baseline)
        try:
            kfc = fc_stack_record(device)
        except Exception as e:   # (a side record must not cost the line)
            kfc = dict(error=str(e)[:300])

    if rank == 0:
        line = {
            "metric": "Mega-pixel-disparities/sec (cost-vol+CBCA+SGM) + end-to-end disp ms/pair",
            "value": round(value, 1), "unit": "MPix-disp/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "ms_per_step_min": round(min(blocks) / args.steps * 1e3, 4), "timed_blocks": len(blocks), "timed_seconds": round(sum(blocks), 3),
            "timing": "blocks of exactly %d steps between barrier + synchronise, repeated until >= %.1f s; value / ms_per_step = median block" % (args.steps, args.min_seconds),
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "pair": PAIR_NOTE[pair],
            "config": {"workload": cfg_name, "H": H, "W": W, "disp_max": D, "feature_channels": abs(C),
                       "params": preset_name, "pairs_per_step": world, "parallelism": "one pair per GPU",
                       "end_to_end_ms_per_pair": round(ms_per_step, 4),
                       "north_star_record": ("the configuration BASELINE.json's north_star target is quoted on (1000x1500x256 accurate, SGM + "
                                             "cross-aggregation sweep) is the `north_star` sub-record of this line: specified texture, "
                                             "`realistic_pair`, `realistic_pair_sample` and `mixed_pair`; KITTI accurate is `kitti_accurate`, its FC stack `kitti_slow_fc`") if north else None},
            "stage_ms": stage, "roofline": roof, "cpu_baseline": cpu, "verify": verify, "ops_ms_per_pair": ops_ms,
            "multi_gpu": multi, "kitti_fast_from_images": fimg, "pipelined": pipe, "kitti_accurate": kacc, "kitti_slow_fc": kfc, "north_star": north,
        }
        print(json.dumps(line))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
