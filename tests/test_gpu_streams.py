"""-m gpu: several pairs in flight.  mc_predict is re-entrant (caller-owned workspace, no state of its own), the host mirror keeps one scratch
area per stream: K pairs queued round-robin on K streams -- the `pipelined` record of bench.py, predict_kitti.py's slots -- must produce, bit
for bit, what each pair produces alone (the reference runs one pair per process, predict_kitti.lua:14-16,61)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("preset,from_images", [("kitti_fast", False), ("kitti_fast", True), ("kitti_slow", False), ("mb_slow", False)])
def test_pairs_in_flight_reproduce_the_pairs_alone(mc, preset, from_images):
    from mc_cnn_amd import main as mcmain
    from mc_cnn_amd.predict import Workspace
    from util import features, natural_pair, raw_volumes, same_bits
    H, W, D, K = 40, 150, 24, 3
    prm = dict(mc.PRESETS[preset])
    dev = torch.device("cuda", 0)
    layers = mcmain.device_layers(mcmain.load_net("random:5", "kitti", "fast"), dev) if from_images else None
    if from_images:
        prm["border_n"] = len(layers)
    slots = []
    for k in range(K):
        x0, x1 = natural_pair(H, W, D, seed=50 + k)
        sl = dict(xb=torch.from_numpy(np.stack([x0, x1])[:, None]).to(dev), ws=Workspace(prm, D, H, W, dev),
                  out=torch.empty((1, 1, H, W), dtype=torch.float32, device=dev), stream=torch.cuda.Stream(device=dev))
        if from_images:
            sl["kw"] = {}
        elif prm["cbca_i1"] + prm["cbca_i2"] == 0:
            sl["kw"] = dict(feat=torch.from_numpy(features(16, H, W, seed=60 + k)).to(dev))
        else:
            raw = raw_volumes(D, H, W, seed=70 + k)
            sl["kw"] = dict(raw=(torch.from_numpy(raw[0]).to(dev), torch.from_numpy(raw[1]).to(dev)))
        slots.append(sl)

    def one(sl):
        kw = dict(feat=mcmain.features_fast(sl["xb"], layers)) if from_images else sl["kw"]
        mc.stereo_predict_fused(sl["xb"], prm, D, workspace=sl["ws"], out=sl["out"], **kw)
    alone = []
    for sl in slots:
        one(sl)
        torch.cuda.synchronize()
        alone.append(sl["out"].cpu().numpy().copy())
        sl["out"].fill_(-1.0)
    torch.cuda.synchronize()
    for rnd in range(3):   # nothing waits in between: three rounds of K pairs queued back to back on their streams
        for sl in slots:
            with torch.cuda.stream(sl["stream"]):
                one(sl)
    torch.cuda.synchronize()
    for k, sl in enumerate(slots):
        assert same_bits(sl["out"].cpu().numpy(), alone[k]), "slot %d differs from the pair run alone" % k
