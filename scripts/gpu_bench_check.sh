#!/bin/bash
# usage (ON the GPU box via gpurun): bash scripts/gpu_bench_check.sh <tag>  -- the driver's bench command, the 2-rank plumbing run, the copy-ceiling table
ulimit -c 0
TAG=${1:-bcheck}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
python - <<PY
import json
try:
    j=json.load(open("$O/bench_default.json"))
    print("default:", j["value"], j["ms_per_step"], j["ms_per_step_min"], j["timed_blocks"], j["timed_seconds"], j["stage_ms"], j["roofline"]["frac"], j["verify"]["bit_exact"])
    k=j["kitti_accurate"]; print("kitti_accurate:", k["ms_per_pair"], k["stage_ms"], k["verify"]["bit_exact"])
    n=j["north_star"]; print("north_star texture:", n["ms_per_pair"], n["stage_ms"], n["per_volume"], n["verify"]["bit_exact"], n.get("cpu_baseline"))
    for r in ("realistic_pair","realistic_pair_sample"):
        print(r, n[r]["ms_per_pair"], n[r]["cbca_ms_per_launch"], n[r]["verify"]["bit_exact"])
except Exception as e:
    print("default bench failed", e); print(open("$O/bench_default.err").read()[-3000:])
PY
MC_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_2ranks.json 2> $O/bench_2ranks.err
python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench_2ranks.json") if l.startswith("{")][-1]); print("2 ranks:", j["config"]["workload"], j["value"], j["ms_per_step"], j["timed_blocks"], j["multi_gpu"])
except Exception as e:
    print("2-rank bench failed", e); print(open("$O/bench_2ranks.err").read()[-3000:])
PY
timeout 200 scripts/microbench/bw_sizes.bin > $O/bw_sizes.txt 2>&1; cat $O/bw_sizes.txt
