"""CPU, world_size 2, gloo: the multi-GPU batch driver (mc-cnn_amd/batch.py) -- pair i -> rank i % world,
one all-gather of the finished disparity maps.  The per-pair computation is a stand-in here (no GPU);
what is tested is the sharding, slot layout and reassembly, including a ragged pair count."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_predict(x_batch):
    # deterministic function of the pair, shape (1,1,H,W)
    return (x_batch[0] * 2 - x_batch[1])[None]


def _load(i, H=6, W=10):
    g = torch.Generator().manual_seed(1000 + i)
    return (torch.randn((2, 1, H, W), generator=g),)


def _worker(rank, world, port, n_pairs, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mc_cnn_amd import batch
        seen = []

        def load(i):
            seen.append(i)
            return _load(i)
        out = batch.predict_pairs(n_pairs, load, _fake_predict, 6, 10, torch.device("cpu"))
        q.put((rank, seen, out.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [4, 5, 1])
def test_predict_pairs_world2(n_pairs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([_fake_predict(*_load(i)).numpy().reshape(6, 10) for i in range(n_pairs)])
    for rank, seen, out in res:
        assert seen == list(range(rank, n_pairs, 2)), "rank %d loaded %s" % (rank, seen)
        assert out.shape == (n_pairs, 6, 10)
        assert np.array_equal(out, want), "rank %d reassembled the batch wrongly" % rank


def test_single_process_path():
    from mc_cnn_amd import batch
    out = batch.predict_pairs(3, _load, _fake_predict, 6, 10, torch.device("cpu"))
    want = torch.stack([_fake_predict(*_load(i)).reshape(6, 10) for i in range(3)])
    assert torch.equal(out, want)
    assert batch.shard(7, 3, 1) == [1, 4]
