"""`predict_kitti.lua` on one node of MI355Xs: the 3-pixel error over the KITTI 2012 training pairs (action `test`) or the
disparity maps of the test pairs as 16-bit PNGs for the evaluation server (action `submit`).

The reference spawns `./main.lua kitti fast -a predict ...` once per pair and reads `disp.bin` back
(`predict_kitti.lua:40-43, 53-63`); here the net is loaded once per process and the pairs are SHARDED over the ranks
(pair i -> rank i % world, `batch.shard`), every pair staying on its GPU; the only exchange is one all-reduce of
(error sum, pair count) at the end, or none at all for `submit` (each rank writes its own files).

    python -m mc_cnn_amd.predict_kitti test   [-path data.kitti/unzip] [-net_fname net/net_kitti_fast_-a_train_all.t7]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m mc_cnn_amd.predict_kitti submit

Within a rank `-pairs_in_flight K` (default 2) pairs are in flight: the PNGs of the next pairs are decoded and normalised by a worker
thread while the GPU runs the current ones, and the pairs alternate between K slots (stream, pinned staging buffers, workspace), so a
pair's upload / feature net / mc_predict / download overlap its neighbours' (bench.py's `pipelined` record: 2.63 -> 2.3-2.5 ms per pair
on the GPU side at KITTI size).  Results do not depend on K.

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


def run(action, path, predict_pair, world=1, rank=0, out_dir="out", n_pairs=None, log=print, in_flight=1):
    """predict_pair(left_png, right_png) -> (H,W) float32 disparity.  Returns (error sum, pairs done) of this rank.
    A predict_pair object with .submit(left_png, right_png) -> ticket and .result(ticket) -> disparity is driven with up to
    `in_flight` pairs submitted ahead of the one whose result is being scored / written (same order, same results)."""
    n = N_PAIRS[action] if n_pairs is None else n_pairs
    err_sum, done = 0.0, 0
    todo = list(shard(n, world, rank))
    staged = hasattr(predict_pair, "submit") and in_flight > 1
    tickets = {}
    ahead = 0
    for k, i in enumerate(todo):
        if staged:
            while ahead < len(todo) and ahead < k + in_flight:
                tickets[todo[ahead]] = predict_pair.submit(*pair_paths(path, action, todo[ahead]))
                ahead += 1
            disp = np.asarray(predict_pair.result(tickets.pop(i)), np.float32)
        else:
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


class PairPipeline:
    """K pairs in flight on one GPU: a pair is handled start to finish by a worker thread on its slot (stream, pinned staging buffers,
    workspace) -- decode + normalise (host side of main.lua:1085-1096), upload, feature net, mc_predict, download -- so that the host
    work of one pair and the GPU work of another overlap, and so do the kernels of two pairs (different streams).  The library is
    re-entrant: one workspace and one scratch area per stream."""

    def __init__(self, prm, layers, disp_max, dev, slots=2):
        import concurrent.futures
        import torch
        self.prm, self.layers, self.disp_max, self.dev = prm, layers, disp_max, dev
        self.slots = [dict(stream=torch.cuda.Stream(device=dev), shape=None) for _ in range(max(1, slots))]
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=len(self.slots))
        self.next = 0

    @staticmethod
    def stage(im0, im1):
        from . import main as mcmain
        x0, x1 = mcmain.load_image(im0), mcmain.load_image(im1)
        if x0.shape[0] == 3:
            x0, x1 = mcmain.rgb2y(x0), mcmain.rgb2y(x1)
        return np.stack([mcmain.normalize(x0), mcmain.normalize(x1)]).astype(np.float32)

    def _work(self, sl, im0, im1):
        import torch
        from . import main as mcmain
        from .predict import Workspace, stereo_predict_fused
        torch.cuda.set_device(self.dev)
        host = self.stage(im0, im1)
        if sl["shape"] != host.shape:   # (KITTI pairs come in a few sizes)
            H, W = host.shape[-2:]
            sl.update(shape=host.shape, pin_in=torch.empty(host.shape, dtype=torch.float32).pin_memory(),
                      pin_out=torch.empty((H, W), dtype=torch.float32).pin_memory(),
                      ws=Workspace(self.prm, self.disp_max, H, W, self.dev))
        sl["pin_in"].copy_(torch.from_numpy(host))
        with torch.cuda.stream(sl["stream"]):
            xb = sl["pin_in"].to(self.dev, non_blocking=True)
            res = stereo_predict_fused(xb, self.prm, self.disp_max, feat=mcmain.features_fast(xb, self.layers), workspace=sl["ws"])
            sl["pin_out"].copy_(res["disp"].reshape(xb.shape[2:]), non_blocking=True)
        sl["stream"].synchronize()
        return sl["pin_out"].numpy().copy()

    def submit(self, im0, im1):
        """Queue a pair on the next slot.  The caller takes results in submission order and keeps at most len(slots) pairs submitted
        (run() does), so a slot's previous pair has been collected before the slot is used again."""
        sl = self.slots[self.next % len(self.slots)]
        self.next += 1
        return self.pool.submit(self._work, sl, im0, im1)

    def result(self, ticket):
        return ticket.result()

    def __call__(self, im0, im1):
        return self.result(self.submit(im0, im1))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="predict_kitti")
    ap.add_argument("action", choices=["test", "submit"])
    ap.add_argument("-path", default="data.kitti/unzip")
    ap.add_argument("-net_fname", default="net/net_kitti_fast_-a_train_all.t7")
    ap.add_argument("-disp_max", type=int, default=228)
    ap.add_argument("-n", type=int, default=None, help="number of pairs (default: the reference's 194 / 195)")
    ap.add_argument("-pairs_in_flight", type=int, default=2, help="pairs staged / queued ahead per rank (1: one at a time)")
    opt = ap.parse_args(argv)
    import torch
    import torch.distributed as dist
    from . import main as mcmain
    from .params import TABLES
    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
    layers = mcmain.device_layers(mcmain.load_net(opt.net_fname, "kitti", "fast"), dev)   # resident: uploaded once
    prm = dict(TABLES[("kitti", "fast")])
    prm["border_n"] = len(layers)
    predict_pair = PairPipeline(prm, layers, opt.disp_max, dev, slots=opt.pairs_in_flight)
    err_sum, done = run(opt.action, opt.path, predict_pair, world, rank, n_pairs=opt.n, in_flight=opt.pairs_in_flight)
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
