"""Debug aid: fused SGM sweeps vs the separate sweeps (MC_SGM_FUSED=0 in a subprocess) on a few shapes."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def run(H, W, D, C=8):
    import torch
    import mc_cnn_amd as mc
    from util import features, smooth_pair
    prm = dict(mc.PRESETS["kitti_fast"])
    x0, x1 = smooth_pair(H, W, min(D, 8), seed=5)
    f = features(C, H, W, seed=6)
    xb = torch.from_numpy(np.stack([x0, x1])[:, None]).cuda()
    r = mc.stereo_predict_fused(xb, prm, D, feat=torch.from_numpy(f).cuda(), want_volumes=True, want_disp0=True)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in r.items()}

if __name__ == "__main__":
    shapes = [(1, 64, 16), (2, 64, 16), (4, 64, 16), (8, 64, 16), (9, 64, 16), (17, 64, 16), (8, 200, 64), (20, 300, 228)]
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        H, W, D = map(int, sys.argv[2:5])
        np.savez("/tmp/sgf_ref_%d_%d_%d.npz" % (H, W, D), **run(H, W, D))
        sys.exit(0)
    for (H, W, D) in shapes:
        env = dict(os.environ, MC_SGM_FUSED="0")
        subprocess.check_call([sys.executable, __file__, "child", str(H), str(W), str(D)], env=env)
        ref = dict(np.load("/tmp/sgf_ref_%d_%d_%d.npz" % (H, W, D)))
        for rep in range(2):
            got = run(H, W, D)
            msg = []
            for k in ("volL", "volR", "dispL0", "disp"):
                a, b = got[k].ravel(), ref[k].ravel()
                bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
                if bad.any():
                    idx = np.flatnonzero(bad)
                    if k.startswith("vol"):
                        d, rem = np.divmod(idx, H * W); yy, xx = np.divmod(rem, W)
                        msg.append("%s: %d bad; y in %s x in [%d,%d] d in [%d,%d]" % (k, bad.sum(), sorted(set(yy.tolist()))[:12], xx.min(), xx.max(), d.min(), d.max()))
                    else:
                        msg.append("%s: %d bad" % (k, bad.sum()))
            print((H, W, D), "rep", rep, "OK" if not msg else "; ".join(msg))
