"""CPU: the C-ABI library loads without a GPU, exports every symbol include/mc_adcensus.h declares,
and validates arguments before touching the device (bad arguments return MC_EINVAL with a message,
where the reference raises a Lua error)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "mc_adcensus.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(mc):
    names = header_symbols()
    assert len(names) >= 27
    lib = C.CDLL(mc._lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libmcadcensus.so does not export %s" % n
    assert sorted(mc._lib.SYMBOLS) == names, "ctypes binding and header disagree"


def test_only_the_abi_is_exported(mc):
    """-fvisibility=hidden: nothing but mc_* leaves the library (no C++ or torch types in the ABI)."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", mc._lib.LIB_PATH]).decode()
    syms = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert syms and all(s.startswith("mc_") for s in syms), syms


def test_version_and_struct_layout(mc):
    assert mc._lib.lib.mc_version() == 8
    # mc_params is plain C: 4-byte fields, one double (8-aligned)
    assert C.sizeof(mc.params.McParams) == 88


def test_bad_arguments_fail_loudly_without_a_gpu(mc):
    lib = mc._lib.lib
    EINVAL = -22
    assert lib.mc_median2d(1, 2, 8, 8, 4, None) == EINVAL           # even kernel size (adcensus.cu:1601)
    assert b"odd" in lib.mc_last_error()
    assert lib.mc_cbca(1, 1, 1, 1, 4, 8, 8, -1, None) == EINVAL       # in == out
    assert lib.mc_sgm2(1, 1, 1, 2, 1, 0, 8, 8, 600, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, -1, None) == EINVAL  # D > 512
    assert lib.mc_sgm2(1, 1, 1, 2, 1, 0, 8, 8, 16, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, -1, None) == EINVAL   # tmp too small
    assert lib.mc_stereo_join(1, 1, 1, 1, 200, 4, 8, 8, None) == EINVAL  # C > 128 (adcensus.cu:1460)
    assert lib.mc_cross(None, 1, 8, 8, 1, 0.5, None) == EINVAL
    assert lib.mc_ad(1, 1, 1, 4, 8, 8, 0, None) == EINVAL              # direction must be -1 / +1
    p = mc.make_params("kitti_fast")
    assert lib.mc_predict_workspace_bytes(C.byref(p), 64, 228, 370, 1226) > 4 * 4 * 228 * 370 * 1226
    assert lib.mc_predict(C.byref(p), 1, 1, None, None, 0, None, None, 8, 8, 8, 256, 1 << 30, None, None, None, None, 1,
                          None) == EINVAL                              # neither features nor raw volumes


def test_plan_and_workspace_sizes(mc):
    """the plan the tile kernel keeps between the aggregation passes of a pair (cbca_tile.hip): item tables of 1 040 bytes per
    (plane, region, step) and 3 bytes per voxel in rows of whole 128-pixel tiles; mc_predict's workspace holds two of them for
    accurate parameter sets with L1 <= 14 and none where nothing would read them"""
    lib = mc._lib.lib
    assert lib.mc_cbca_plan_bytes(0, 8, 8) == 0
    for D, H, W in ((256, 1000, 1500), (228, 370, 1226), (7, 33, 130)):
        gx = -(-W // 128)
        rows = 3 * D * H * gx * 128
        got = lib.mc_cbca_plan_bytes(D, H, W)
        steps = -(-H // 16)
        assert rows + 1040 * D * gx * steps <= got <= rows + 1040 * D * gx * (steps + 8) + 1024, (D, H, W, got)
    assert abs(lib.mc_cbca_plan_bytes(256, 1000, 1500) / (256 * 1000 * 1500) - 3.6) < 0.1   # bytes per voxel
    fast, slow, mb = mc.make_params("kitti_fast"), mc.make_params("kitti_slow"), mc.make_params("mb_slow")
    V = 4 * 228 * 370 * 1226
    ws_fast = lib.mc_predict_workspace_bytes(C.byref(fast), 64, 228, 370, 1226)
    ws_slow = lib.mc_predict_workspace_bytes(C.byref(slow), 0, 228, 370, 1226)
    assert 6 * V < ws_fast < 6.3 * V
    assert ws_slow - ws_fast >= 2 * lib.mc_cbca_plan_bytes(228, 370, 1226) - (1 << 20)          # two plans (and the packed arms)
    assert ws_slow - ws_fast < 2 * lib.mc_cbca_plan_bytes(228, 370, 1226) + 64 * 370 * 1226
    once = mc.make_params("kitti_slow"); once.cbca_i1, once.cbca_i2 = 1, 0     # a single pass per direction: nothing to keep
    assert lib.mc_predict_workspace_bytes(C.byref(once), 0, 228, 370, 1226) < ws_fast + 64 * 370 * 1226
    long_arms = mc.make_params("mb_slow"); long_arms.L1 = 20                     # strip kernel only: no plan
    assert lib.mc_predict_workspace_bytes(C.byref(mb), 0, 256, 1000, 1500) - lib.mc_predict_workspace_bytes(C.byref(long_arms), 0, 256, 1000, 1500) >= \
        2 * lib.mc_cbca_plan_bytes(256, 1000, 1500)


def test_gaussian_host_matches_main_lua(mc):
    """gaussian(sigma), main.lua:528-540, runs on the host in doubles: checked here without a GPU."""
    import math
    import numpy as np
    k = mc.adcensus.gaussian(1.67).numpy()
    kr = math.ceil(1.67 * 3)
    assert k.shape == (2 * kr + 1, 2 * kr + 1)
    want = np.array([[math.exp(-(x * x + y * y) / (2 * 1.67 * 1.67)) for x in range(-kr, kr + 1)]
                     for y in range(-kr, kr + 1)]).astype(np.float32)
    assert (k == want).all()


def test_product_never_imports_the_oracle():
    """The product path must not route through the CPU oracle (tests / smoke / bench baseline only)."""
    pkg = os.path.join(ROOT, "mc-cnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".lua")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "cpu_oracle" not in txt and "mc_oracle" not in txt and "ref_lib" not in txt, f
