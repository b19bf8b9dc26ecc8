#!/bin/bash
# usage (ON the GPU box via gpurun): bash scripts/gpu_roll.sh <tag> [all]  -- the rolling form of the tile kernel's plan-reading pass
ulimit -c 0
TAG=${1:-roll}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
MC_REQUIRE_REF=1 timeout 300 python -m pytest tests/test_gpu_cbca_tile.py -x -q -k "with_plan" > $O/pytest_roll.log 2>&1; echo "pytest roll rc=$?"; tail -3 $O/pytest_roll.log | grep -v amdgpu.ids
timeout 300 python scripts/gpu_cbca_tile.py 14natural 5natural --plan-only 2>&1 | grep -v amdgpu.ids | tee $O/roll_times.txt
if [ "$2" = all ]; then MC_REQUIRE_REF=1 timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $O/pytest_all.log | grep -v amdgpu.ids; fi
