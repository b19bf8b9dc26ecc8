"""-m gpu: the N>1 path of bench.py with the REAL mc_predict on the one GPU of the test box: world_size 2, both
ranks on cuda:0 (MC_BENCH_ONE_GPU=1: collectives over gloo), ranks spawned by bench.py itself because WORLD_SIZE
is unset -- the call the driver may make (`python bench.py --gpus N`).  Checks the JSON line: both ranks took part,
every rank found its own map in its slot of the gathered batch, and the maps rank 0 gathered equal what a
single-GPU run produces for those pairs (predict_kitti.lua:61: one independent process per pair)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config", ["tiny", "kitti_fast"])
def test_bench_two_ranks_one_device(config):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MC_BENCH_ONE_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--config", config], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["pairs_per_step"] == 2
    m = j["multi_gpu"]
    assert m["ranks_seen"] == [0, 1] and m["world_size"] == 2
    assert m["own_slot_bit_exact_all_ranks"] is True
    assert m["ranks_recomputed_on_rank0"] == [1] and m["gathered_equals_single_gpu"] is True
    assert j["value"] > 0 and j["roofline"]["frac"] > 0
