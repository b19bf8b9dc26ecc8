"""-m gpu: which kernel mc_predict's aggregation passes take where it keeps a plan area and knows 5 <= L1 <= 14 (round 5).

The plan area of a direction belongs to ONE of two users, decided on the device: the tile kernel's plan (real-scene arms) or the
texture route's records.  The records are classified before the direction's first pass; from then on a pass is the tile kernel's
long-arm instance (the tile routes, and a texture whose records are unusable: flat regions next to it) or the texture route's own
kernels (cbca_lean2x per pair of passes, the strip kernel for a single one).  Checked here, against the oracle and through the two
heads of the area (list: words 0..7, plan: words 32..37):

* a single first aggregation pass followed by pairs (cbca_i1 = 1, cbca_i2 >= 2) on real-scene arms -- the plan the
  first pass wrote must survive the second stage's classification (it was zeroed by an unconditional memset: every later tile pass
  stood down and the volumes were stale);
* the regime between the two (tests/util.mixed_pair): a texture with flat patches -> route CR_STRIP, records overflow -> tile kernel;
* one workspace reused across pairs of different routes (a stale head of the other user must never be trusted)."""
import numpy as np
import pytest

from util import diff_report, mixed_pair, natural_pair, raw_volumes, same_bits, smooth_pair

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

LIST_MAGIC, PLAN_MAGIC = 0x4c495354, 0x504c414e


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def heads(mc, ws, D, H, W, nplan=2):
    """(list head words 0..7, plan head words 32..37) of every direction's plan area at the end of the workspace"""
    lib = mc._lib.lib
    cplan = (lib.mc_cbca_plan_bytes(D, H, W) + 255) // 256 * 256
    off = ws.ptr - ws.buf.data_ptr()
    out = []
    for v in range(nplan):
        a = off + ws.nbytes - (nplan - v) * cplan
        words = ws.buf[a:a + 256].cpu().numpy().view(np.uint32)
        out.append((words[0:8].tolist(), words[32:38].tolist()))
    return out


def run(mc, oracle, prm, x0, x1, D, vl, vr, ws=None):
    H, W = x0.shape
    with np.errstate(all="ignore"):
        want = oracle.stereo_predict(prm, x0, x1, D, rawL=vl, rawR=vr)
    xb = dev(np.stack([x0, x1]))[:, None]
    ws = ws or mc.predict.Workspace(prm, D, H, W, xb.device)
    got = mc.stereo_predict_fused(xb, prm, D, raw=(dev(vl), dev(vr)), workspace=ws, want_volumes=True, want_disp0=True)
    torch.cuda.synchronize()
    for k in ("volL", "volR", "dispL0", "dispR0", "disp"):
        g = got[k].cpu().numpy()
        assert same_bits(g, want[k]), diff_report(g, want[k], k)
    return ws


@pytest.mark.parametrize("i1,i2", [(1, 2), (1, 5), (1, 3), (3, 2), (0, 3)])
def test_single_first_pass_then_pairs_on_real_scene_arms(mc, oracle, i1, i2):
    H, W, D = 60, 300, 16
    prm = dict(mc.PRESETS["mb_slow"], cbca_i1=i1, cbca_i2=i2)
    x0, x1 = natural_pair(H, W, 8, seed=4, sigma=8.0)
    vl, vr = raw_volumes(D, H, W, seed=7)
    ws = run(mc, oracle, prm, x0, x1, D, vl, vr)
    for lst, pln in heads(mc, ws, D, H, W):
        assert pln[0] == PLAN_MAGIC and pln[1:4] == [D, H, W] and pln[5] == 13, "the tile kernel's plan is not in place: %r" % (pln,)


@pytest.mark.parametrize("i1,i2", [(2, 3), (1, 2), (2, 0), (1, 4)])
def test_texture_with_flat_patches_goes_to_the_tile_kernel(mc, oracle, i1, i2):
    H, W, D = 96, 520, 16
    prm = dict(mc.PRESETS["mb_slow"], cbca_i1=i1, cbca_i2=i2)
    x0, x1 = mixed_pair(H, W, D, seed=5, flat_frac=0.04, patch=44)   # 80 % unit-arm pixels: the texture route; patches of >= 22 x 44 flat pixels
    vl, vr = raw_volumes(D, H, W, seed=7)
    ws = run(mc, oracle, prm, x0, x1, D, vl, vr)
    for lst, pln in heads(mc, ws, D, H, W):
        assert lst[6] == LIST_MAGIC and lst[1] != 0, "the records of a texture with flat patches should have been declared unusable: %r" % (lst,)
        assert pln[0] == PLAN_MAGIC and pln[5] == 13, "... and the tile kernel should have taken the passes: %r" % (pln,)


def test_texture_keeps_its_own_kernels(mc, oracle):
    H, W, D = 70, 420, 24
    prm = dict(mc.PRESETS["mb_slow"], cbca_i1=1, cbca_i2=4)   # a single pass (strip kernel, the list already classified), then two pairs
    x0, x1 = smooth_pair(H, W, 10, seed=9)
    vl, vr = raw_volumes(D, H, W, seed=5)
    ws = run(mc, oracle, prm, x0, x1, D, vl, vr)
    for lst, pln in heads(mc, ws, D, H, W):
        assert lst[6] == LIST_MAGIC and lst[1] == 0 and lst[2:5] == [D, H, W], "the texture's records: %r" % (lst,)
        assert pln[0] == 0, "no tile pass should have run on a texture: %r" % (pln,)


def test_one_workspace_across_routes(mc, oracle):
    """natural -> texture -> mixed -> natural on ONE workspace: every call finds the other user's head in the area"""
    H, W, D = 96, 520, 16
    prm = dict(mc.PRESETS["mb_slow"], cbca_i1=2, cbca_i2=3)
    vl, vr = raw_volumes(D, H, W, seed=11)
    ws = None
    pairs = [natural_pair(H, W, 8, seed=4, sigma=8.0), smooth_pair(H, W, 10, seed=9), mixed_pair(H, W, D, seed=5, flat_frac=0.04, patch=44),
             natural_pair(H, W, 8, seed=6, sigma=8.0), smooth_pair(H, W, 10, seed=2)]
    for x0, x1 in pairs:
        ws = run(mc, oracle, prm, x0, x1, D, vl, vr, ws)


def test_arms_of_at_most_four_under_the_long_arm_parameters(mc, oracle):
    """L1 = 14 but no arm beyond 4 in the pair (route CR_TILE4): served by the long-arm instance in the planned passes"""
    H, W, D = 48, 200, 8
    prm = dict(mc.PRESETS["mb_slow"], cbca_i1=2, cbca_i2=2, tau1=0.3)
    rng = np.random.default_rng(3)
    # blocks of 4 x 4 equal pixels with large steps between them: arms of 1 .. 4, few unit-arm pixels (not a texture)
    def img():
        small = rng.integers(0, 50, size=(H // 4, W // 4)).astype(np.float32)
        return np.kron(small, np.ones((4, 4), np.float32))
    x0, x1 = img(), img()
    vl, vr = raw_volumes(D, H, W, seed=2)
    run(mc, oracle, prm, x0, x1, D, vl, vr)
