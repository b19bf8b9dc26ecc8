#!/bin/bash
# usage (ON the GPU box via gpurun): bash scripts/gpu_lean.sh <tag>  -- the lean path: its tests, the fused path at full size, per-kernel times
ulimit -c 0
TAG=${1:-lean}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
MC_REQUIRE_REF=1 timeout 600 python -m pytest tests/test_gpu_cbca_lean.py -x -q > $O/pytest_lean.log 2>&1; echo "pytest lean rc=$?"; tail -3 $O/pytest_lean.log | grep -v amdgpu.ids
MC_REQUIRE_REF=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "mb" > $O/pytest_full.log 2>&1; echo "pytest fullsize rc=$?"; tail -3 $O/pytest_full.log | grep -v amdgpu.ids
if [ "$2" = all ]; then MC_REQUIRE_REF=1 timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_cbca_lean.py > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $O/pytest_all.log | grep -v amdgpu.ids; fi
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/prof_lean -o lean -- python $GRAFT_REPO_ROOT/scripts/gpu_cbca_lean.py > $O/lean_times.txt 2>&1
cd $GRAFT_REPO_ROOT; grep -v amdgpu.ids $O/lean_times.txt | tail -25
python scripts/rocpd_by_grid.py $O/prof_lean/lean_results.db > $O/lean_by_grid.csv 2>&1; grep -i "lean\|list\|classify\|strip" $O/lean_by_grid.csv | sed "s/mc::(anonymous namespace)::LeanArgs//"
timeout 300 python bench.py --config mb_slow --steps 3 --warmup 2 --no-cpu-baseline --no-ops > $O/bench_mb_slow.json 2> $O/bench_mb_slow.err
python - <<PY
import json
try:
    j=json.load(open("$O/bench_mb_slow.json")); print("mb_slow", j["ms_per_step"], j["stage_ms"], j["roofline"]["frac"], j["verify"]["bit_exact"])
except Exception as e: print("bench failed", e); print(open("$O/bench_mb_slow.err").read()[-1500:])
PY
if [ "$3" = prof ]; then bash scripts/gpu_prof.sh $TAG mb_slow 2; fi
