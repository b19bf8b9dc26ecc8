"""bench.py pieces that need no GPU: the algorithmic-bytes model is SURVEY.md section 8(d)'s, the workload table names
BASELINE.json's configurations, and the default workload is configs[1]."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_follow_survey_8d():
    b = _bench()
    import mc_cnn_amd as mc
    H, W, D, C = 370, 1226, 228, 64
    V, F = 4.0 * D * H * W, 4.0 * C * H * W
    fast = b.algorithmic_bytes(dict(mc.PRESETS["kitti_fast"]), H, W, D, C)
    assert fast["join"] == 2 * F + 2 * V            # read both feature maps, write both volumes
    assert fast["sgm"] == 2 * 11 * V                # 11 V per volume and sgm2 call
    assert fast["cbca"] == 0 and fast["argmin"] == 2 * V
    slow = b.algorithmic_bytes(dict(mc.PRESETS["mb_slow"]), 1000, 1500, 256, 0)
    Vm = 4.0 * 256 * 1000 * 1500
    assert slow["join"] == 0 and slow["cbca"] == 2 * (2 + 16) * 2 * Vm   # 2 V per iteration and volume
    assert abs(slow["total"] - (slow["cbca"] + slow["sgm"] + slow["argmin"])) < 1


def test_workloads_are_baseline_configs():
    b = _bench()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert b.CONFIGS["kitti_fast"][1:4] == (370, 1226, 228) and "KITTI 2012 fast" in base["configs"][1]
    assert b.CONFIGS["kitti_slow"][1:4] == (370, 1226, 228) and "accurate" in base["configs"][2]
    assert b.CONFIGS["mb_slow"][1:4] == (1000, 1500, 256) and "1500x1000" in base["configs"][3]
    assert b.HBM_PEAK_GBS == 8000.0
    src = open(os.path.join(ROOT, "bench.py")).read()
    # BASELINE configs[1] is what `python bench.py` measures on one GPU, configs[4] (a Middlebury-size pair per rank) on several
    assert 'args.config = "kitti_fast" if args.gpus <= 1 else "mb_slow"' in src
    assert base["metric"].startswith("Mega-pixel-disparities/sec")


def test_dominant_kernel_is_picked_by_time_not_bytes():
    """for kitti_slow the byte model is largest for SGM while CBCA takes the most time."""
    b = _bench()
    import mc_cnn_amd as mc
    prm = dict(mc.PRESETS["kitti_slow"])
    ab = b.algorithmic_bytes(prm, 370, 1226, 228, 0)
    assert ab["sgm"] > ab["cbca"]
    assert b.pick_dominant({"cbca": 2.8, "sgm": 2.0, "join": 0.0}, ab) == "cbca"
    assert b.pick_dominant({"cbca": 0.5, "sgm": 2.0}, ab) == "sgm"
    rec = b.roofline_record("kitti_slow", prm, 370, 1226, 228, 0, {"cbca": 2.8, "sgm": 2.0}, 6.0)
    assert rec["kernel"].startswith("cbca_") and set(rec["kernels"]) == {"cbca", "sgm"}
    assert abs(rec["kernels"]["sgm"]["frac"] - ab["sgm"] / 2.0e-3 / 1e9 / 8000.0) < 1e-3
    assert b.launches_per_step(dict(mc.PRESETS["mb_slow"]), 0) == {"join": 0, "cbca": 36, "sgm": 3}


def test_gpus_n_without_a_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus N` must not exit 2 when WORLD_SIZE is unset: it starts the ranks itself."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if args.gpus > 1 and "WORLD_SIZE" not in os.environ:' in src and "spawn_ranks(args.gpus)" in src
    assert "ranks_seen" in src and "gathered_equals_single_gpu" in src
    assert '"verify": verify' in src and '"north_star": north' in src


def test_every_config_names_its_image_pair():
    """the pair decides the cost of cbca: KITTI shapes on the reference's real sample pair, 1000x1500 on the specified texture"""
    b = _bench() if "_bench" in globals() else __import__("bench")
    assert set(b.PAIR_OF) == set(b.CONFIGS)
    assert all(v in b.PAIR_NOTE for v in b.PAIR_OF.values())
    assert b.PAIR_OF["kitti_fast"] == b.PAIR_OF["kitti_slow"] == "sample" and b.PAIR_OF["mb_slow"] == "texture"
    assert b.config_key(b.CONFIGS["kitti_slow_fc"]) == "kitti_slow_fc"


def test_support_sizes_count_the_taps_of_the_reference_loop(oracle):
    """bench's `additions_per_voxel`: the closed form against a literal walk of adcensus.cu:356-373 on the oracle's arms"""
    import numpy as np
    from util import natural_pair
    b = __import__("bench")
    H, W = 30, 70
    x0, x1 = natural_pair(H, W, 8, seed=3, sigma=8.0)
    aL, aR = b.arm_lengths(oracle.cross(x0, 14, 0.05)), b.arm_lengths(oracle.cross(x1, 14, 0.05))
    for d in (0, 5):
        size = b.support_sizes(aL, aR, d)
        m = np.minimum(aL[:, :, d:], aR[:, :, :W - d] if d else aR)
        for y in range(0, H, 3):
            for x in range(0, W - d, 5):
                l, r, u, dn = m[:, y, x]
                want = sum(int(m[0, q, x] + m[1, q, x] + 1) for q in range(y - u, y + dn + 1))
                assert size[y, x] == want


def test_multi_gpu_default_is_configs4_with_per_rank_pairs():
    """`bench.py --gpus N` (N > 1) measures BASELINE configs[4]: one Middlebury-size accurate pair per GPU, every rank its own
    seeded pair -- checked without a GPU through --dry-run (the inputs a rank would process, fingerprinted)"""
    import subprocess, sys
    recs = []
    for r in (0, 1):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                             text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        recs.append(json.loads(out.stdout.strip().splitlines()[-1]))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "Middlebury-size" in base["configs"][4]
    for r, rec in enumerate(recs):
        assert rec["config"] == "mb_slow" and rec["rank"] == r and "1000x1500" in rec["workload"]
        assert rec["seeds"] == {"images": 1234 + r, "raw_volumes": 7 + r}          # configs[4]: seeds 7 ... 14, one per GPU
    for k in ("x0", "x1", "raw"):
        assert recs[0]["inputs"][k] != recs[1]["inputs"][k], "both ranks would process the same %s" % k
    if recs[0]["cpus"] and recs[1]["cpus"]:   # disjoint shares of the host cores
        assert not set(recs[0]["cpus"]) & set(recs[1]["cpus"])
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--config", "tiny"], capture_output=True, text=True, timeout=600)
    assert json.loads(one.stdout.strip().splitlines()[-1])["config"] == "tiny"
    d = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], capture_output=True, text=True, timeout=600)
    assert json.loads(d.stdout.strip().splitlines()[-1])["config"] == "kitti_fast"   # one GPU: BASELINE configs[1]


def test_texture_route_launches_run_two_iterations_each():
    bench = _bench()
    """cbca_lean2x_kernel runs the iterations of a textured pair two per launch (an odd last one on its own): the roofline record's
    launches, algorithmic bytes per launch and traffic per launch follow; the other routes keep one iteration per launch"""
    import mc_cnn_amd as mc
    mb = dict(mc.PRESETS["mb_slow"])
    assert (mb["cbca_i1"], mb["cbca_i2"]) == (2, 16)
    assert bench.launches_per_step(mb, 0, "texture")["cbca"] == 2 * (1 + 8)
    assert bench.launches_per_step(mb, 0, "natural")["cbca"] == 2 * 18 == bench.launches_per_step(mb, 0)["cbca"]
    odd = dict(mb, cbca_i1=1, cbca_i2=5)
    assert bench.launches_per_step(odd, 0, "texture")["cbca"] == 2 * (1 + 3)
    kitti = dict(mc.PRESETS["kitti_slow"])                       # L1 = 5: arms <= 4, the tile kernel's short-arm instance whatever the pair
    assert bench.launches_per_step(kitti, 0, "texture")["cbca"] == 2 * (kitti["cbca_i1"] + kitti["cbca_i2"])
    H, W, D = 1000, 1500, 256
    acc = {"cbca": 16.0, "sgm": 6.4}
    rec = bench.roofline_record("mb_slow", mb, H, W, D, 0, acc, 25.0, None, None, "texture")
    assert rec["launches_per_step"] == 18 and rec["algorithmic_bytes_per_launch"] == 4 * 4 * D * H * W
    assert "cbca_lean2x_kernel" in rec["kernel"] and "one iteration over one volume per launch" not in rec["kernel"]
    tfile = os.path.join(ROOT, "profiles", "traffic_mb_slow.json")
    if os.path.exists(tfile):                                     # profiles/traffic_*.json: cbca's bytes per ITERATION
        per_it = json.load(open(tfile))["cbca"]
        assert rec["traffic"] == round(per_it * 2)
        assert rec["traffic"] < rec["algorithmic_bytes_per_launch"]   # the point of the kernel: fewer bytes moved than the count


def test_traffic_files_say_where_their_numbers_come_from():
    """`roofline.traffic` is a builder-kept constant (a rocprofv3 --pmc pass of the evidence run), and the record must say so: every
    committed profiles/traffic_<config>.json is stamped with the run and the commit it was collected on, and bench.py turns that into `traffic_source`"""
    import glob
    import bench
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "profiles", "traffic_*.json")))
    assert len(files) >= 6
    for f in files:
        key = os.path.basename(f)[len("traffic_"):-len(".json")]
        t, src = bench.traffic_file(key)
        assert t.get("run") and t.get("commit") and t["commit"] != "n/a", f
        assert src.startswith("profiles/traffic_%s.json" % key) and "not measured in this run" in src and t["commit"] in src
    assert bench.traffic_file("no_such_config") == ({}, None)


def test_round6_side_records_are_wired_into_the_default_line():
    """the image -> disparity record (feature net included) and the pairs-in-flight record ride in the default line, ahead of the long
    north_star record; the convolution's roofline is priced against the fp32 matrix peak, on the reference's flop count"""
    import inspect
    b = _bench()
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('"kitti_fast_from_images": fimg') < src.index('"kitti_accurate": kacc') < src.index('"north_star": north')
    f = inspect.getsource(b.from_images_record)
    assert "features_fast" in f and "157.3" in f and "2.0 * 2 * H * W * fm * fm * 9" in f
    assert "conv2d" in f and "verify_against_reference" in f          # checked against a float64 convolution and the reference's kernels
    p = inspect.getsource(b.pipelined_record)
    assert "torch.cuda.Stream" in p and "Workspace(" in p and "same_bits_dev" in p
    assert "--pairs-in-flight" in src


def test_conv_plan_splits_banks_that_do_not_fit():
    """mc_conv3x3's workspace holds the re-laid filter bank: groups x pairs (even) x 64 lanes x 9 taps x tiles of 32 channels"""
    import mc_cnn_amd as mc
    lib = mc._lib.lib
    assert lib.mc_conv3x3_workspace_bytes(64, 64) == 32 * 64 * 9 * 2 * 4            # one group of 64 output channels: 144 KB, resident in LDS
    assert lib.mc_conv3x3_workspace_bytes(1, 64) == 2 * 64 * 9 * 2 * 4              # the pair count is padded to even
    assert lib.mc_conv3x3_workspace_bytes(112, 112) == 4 * 56 * 64 * 9 * 1 * 4      # four groups of 32 output channels (126 KB each)
    assert lib.mc_conv3x3_workspace_bytes(0, 64) == 0 and lib.mc_conv3x3_workspace_bytes(64, 129) == 0
