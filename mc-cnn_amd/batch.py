"""A batch of stereo pairs across the GPUs of one node: one process per GPU, pair i -> rank i % world,
no data-path collective while the pairs are processed, ONE gather of the finished disparity maps at
the end (RCCL over xGMI on the GPU box; `nccl` backend == RCCL on ROCm).

The reference has nothing to mirror here: it handles one pair per process (`predict_kitti.lua:61`
spawns `main.lua -a predict` per image pair) and its only fan-out is process-level (`rgs.py:79-91`).
Cost volumes (V = 4*D*H*W bytes each) never leave their GPU; only H*W*4 bytes per pair are exchanged.
"""
import torch
import torch.distributed as dist


def shard(n_pairs, world, rank):
    """Indices of the pairs rank `rank` owns: round-robin, so consecutive pairs run concurrently."""
    return list(range(rank, n_pairs, world))


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def predict_pairs(n_pairs, load_pair, predict_fn, H, W, device, group=None):
    """Run `predict_fn(*load_pair(i))` -> (1,1,H,W) disparity on this rank's shard of pairs 0..n_pairs-1
    and return, on EVERY rank, the (n_pairs,H,W) tensor of all disparity maps.

    load_pair(i) is only called for pairs this rank owns (inputs stay device-local).  The one
    collective is an all-gather of a (slots,H,W) block per rank, slots = ceil(n_pairs / world);
    unused slots are NaN and dropped on reassembly.
    """
    world, rank = _world(group)
    slots = (n_pairs + world - 1) // world
    mine = shard(n_pairs, world, rank)
    local = torch.full((slots, H, W), float("nan"), dtype=torch.float32, device=device)
    for k, i in enumerate(mine):
        out = predict_fn(*load_pair(i))
        local[k].copy_(out.reshape(H, W))
    if world == 1:
        return local[:n_pairs]
    gathered = torch.empty((world, slots, H, W), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(gathered.view(world * slots, H, W), local, group=group)
    # pair i lives at [i % world, i // world]
    idx = torch.arange(n_pairs, device=device)
    return gathered[idx % world, idx // world]
