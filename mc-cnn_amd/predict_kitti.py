"""`predict_kitti.lua` on one node of MI355Xs: the 3-pixel error over the KITTI 2012 training pairs (action `test`) or the
disparity maps of the test pairs as 16-bit PNGs for the evaluation server (action `submit`).

The reference spawns `./main.lua kitti fast -a predict ...` once per pair and reads `disp.bin` back
(`predict_kitti.lua:40-43, 53-63`); here the net is loaded once per process and the pairs are SHARDED over the ranks
(pair i -> rank i % world, `batch.shard`), every pair staying on its GPU; the only exchange is one all-reduce of
(error sum, pair count) at the end, or none at all for `submit` (each rank writes its own files).

    python -m mc_cnn_amd.predict_kitti test   [-path data.kitti/unzip] [-net_fname net/net_kitti_fast_-a_train_all.t7]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m mc_cnn_amd.predict_kitti submit

Same file layout as the reference: <path>/{training,testing}/image_{0,1}/%06d_10.png, ground truth
<path>/training/disp_noc/%06d_10.png (PNG16, value/256, 0 = no ground truth), output out/%06d_10.png.
"""
import argparse
import os
import sys

import numpy as np

from .batch import shard
from .binio import read_png16, write_png16

N_PAIRS = {"test": 194, "submit": 195}   # predict_kitti.lua:47-52


def pair_paths(path, action, i):
    d = "training" if action == "test" else "testing"
    return ("%s/%s/image_0/%06d_10.png" % (path, d, i), "%s/%s/image_1/%06d_10.png" % (path, d, i))


def three_pixel_error(disp, ground_truth):
    """predict_kitti.lua:70-73: share of the pixels WITH ground truth whose |disp - gt| exceeds 3."""
    mask = ground_truth != 0
    bad = (np.abs(disp - ground_truth) > 3) & mask
    return float(bad.sum()) / float(mask.sum())


def run(action, path, predict_pair, world=1, rank=0, out_dir="out", n_pairs=None, log=print):
    """predict_pair(left_png, right_png) -> (H,W) float32 disparity.  Returns (error sum, pairs done) of this rank."""
    n = N_PAIRS[action] if n_pairs is None else n_pairs
    err_sum, done = 0.0, 0
    for i in shard(n, world, rank):
        im0, im1 = pair_paths(path, action, i)
        disp = np.asarray(predict_pair(im0, im1), np.float32)
        if action == "test":
            gt = read_png16("%s/training/disp_noc/%06d_10.png" % (path, i))
            err = three_pixel_error(disp, gt)
            err_sum += err
            log(i, err)
        else:
            os.makedirs(out_dir, exist_ok=True)
            write_png16(disp, "%s/%06d_10.png" % (out_dir, i))
            log(i)
        done += 1
    return err_sum, done


def main(argv=None):
    ap = argparse.ArgumentParser(prog="predict_kitti")
    ap.add_argument("action", choices=["test", "submit"])
    ap.add_argument("-path", default="data.kitti/unzip")
    ap.add_argument("-net_fname", default="net/net_kitti_fast_-a_train_all.t7")
    ap.add_argument("-disp_max", type=int, default=228)
    ap.add_argument("-n", type=int, default=None, help="number of pairs (default: the reference's 194 / 195)")
    opt = ap.parse_args(argv)
    import torch
    import torch.distributed as dist
    from . import main as mcmain
    from .params import TABLES
    from .predict import stereo_predict_fused
    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
    layers = mcmain.load_net(opt.net_fname, "kitti", "fast")
    prm = dict(TABLES[("kitti", "fast")])
    prm["border_n"] = len(layers)

    # one HIP stream, one pinned staging buffer each way and one workspace per rank: uploads, the pipeline and the download of
    # a pair are queued on the rank's own stream (the C ABI launches on torch's current stream) and the host waits once per pair
    from .predict import Workspace
    stream = torch.cuda.Stream(device=dev)
    st = {"shape": None}

    def predict_pair(im0, im1):
        x0, x1 = mcmain.load_image(im0), mcmain.load_image(im1)
        if x0.shape[0] == 3:
            x0, x1 = mcmain.rgb2y(x0), mcmain.rgb2y(x1)
        host = np.stack([mcmain.normalize(x0), mcmain.normalize(x1)]).astype(np.float32)
        if st["shape"] != host.shape:   # (KITTI pairs come in a few sizes)
            H, W = host.shape[-2:]
            st.update(shape=host.shape, pin_in=torch.empty(host.shape, dtype=torch.float32).pin_memory(),
                      pin_out=torch.empty((H, W), dtype=torch.float32).pin_memory(),
                      ws=Workspace(prm, opt.disp_max, H, W, dev))
        st["pin_in"].copy_(torch.from_numpy(host))
        with torch.cuda.stream(stream):
            xb = st["pin_in"].to(dev, non_blocking=True)
            res = stereo_predict_fused(xb, prm, opt.disp_max, feat=mcmain.features_fast(xb, layers), workspace=st["ws"])
            st["pin_out"].copy_(res["disp"].reshape(xb.shape[2:]), non_blocking=True)
        stream.synchronize()
        return st["pin_out"].numpy().copy()

    err_sum, done = run(opt.action, opt.path, predict_pair, world, rank, n_pairs=opt.n)
    if opt.action == "test":
        t = torch.tensor([err_sum, float(done)], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t)
        if rank == 0:
            print(t[0].item() / max(t[1].item(), 1.0))   # predict_kitti.lua:83
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
