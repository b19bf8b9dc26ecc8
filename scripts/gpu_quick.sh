#!/bin/bash
# usage: bash scripts/gpu_quick.sh <tag> [pytest-args...]   (runs ON the GPU box via gpurun)
TAG=${1:-quick}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q "$@" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
for cfg in kitti_fast kitti_slow mb_slow; do
  steps=20; [ $cfg = mb_slow ] && steps=3
  timeout 300 python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-ref-gpu > $O/bench_$cfg.json 2> $O/bench_$cfg.err
  python - <<PY
import json
try:
    j=json.load(open("$O/bench_$cfg.json")); print("$cfg", j["ms_per_step"], j["stage_ms"], j["roofline"]["frac"])
except Exception as e: print("$cfg failed", e); print(open("$O/bench_$cfg.err").read()[-2000:])
PY
done
