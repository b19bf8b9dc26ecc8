#!/bin/bash
# usage (ON the GPU box via gpurun): bash scripts/gpu_lean2x.sh <tag> -- the two-passes-per-launch kernel: its tests (+ the single-pass lean
# tests, whose classification it shares), the textured 1000x1500x256 bench line verified against the reference's kernels, kernel times
ulimit -c 0
TAG=${1:-lean2x}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
MC_REQUIRE_REF=1 timeout 900 python -m pytest tests/test_gpu_cbca_lean2x.py tests/test_gpu_cbca_lean.py -m gpu -q --tb=short --maxfail=12 > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids $O/pytest.log | tail -60 | cut -c1-400
timeout 400 python bench.py --config mb_slow --steps 3 --warmup 1 --no-cpu-baseline --no-ops > $O/bench_mb_slow.json 2> $O/bench_mb_slow.err
python - <<PY
import json
try:
    j=json.load(open("$O/bench_mb_slow.json")); print("mb_slow:", j["ms_per_step"], j["stage_ms"], j["roofline"], j.get("verify"))
except Exception as e: print("mb_slow failed", e); print(open("$O/bench_mb_slow.err").read()[-3000:])
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o mb -- python $GRAFT_REPO_ROOT/bench.py --config mb_slow --steps 2 --warmup 1 --no-cpu-baseline --no-ref-gpu --no-ops > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -14 $f | cut -c1-200
