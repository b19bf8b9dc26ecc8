#!/bin/bash
# round 2 checkpoint: whole -m gpu suite, smoke, default bench line (verify + north_star), 2-rank self-spawn
ulimit -c 0
TAG=${1:-r2d}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
nproc > $O/nproc.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_kitti_fast.json 2> $O/bench_kitti_fast.err; tail -2 $O/bench_kitti_fast.err
python - <<PY
import json
j=json.load(open("$O/bench_kitti_fast.json"))
print("kitti_fast", j["value"], j["ms_per_step"], j["stage_ms"], j["roofline"]["frac"], j["roofline"]["kernels"])
print("verify", j["verify"]); print("ops", j["ops_ms_per_pair"]); print("cpu", j["cpu_baseline"])
n=j["north_star"]; print("north", n["ms_per_pair"], n["per_volume"], n["verify"]["bit_exact"], n["stage_ms"])
PY
for cfg in kitti_slow mb_slow; do
timeout 600 python bench.py --config $cfg --steps 5 --warmup 1 > $O/bench_$cfg.json 2> $O/bench_$cfg.err
python - <<PY
import json
j=json.load(open("$O/bench_$cfg.json"))
print("$cfg", j["value"], j["ms_per_step"], j["stage_ms"], j["roofline"]["kernel"][:20], j["roofline"]["frac"], j["verify"]["bit_exact"], j["ops_ms_per_pair"], j["cpu_baseline"]["value"])
PY
done
