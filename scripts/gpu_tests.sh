#!/bin/bash
# full -m gpu suite (runs ON the GPU box via gpurun): scripts/gpu_tests.sh [pytest args]
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/tests; mkdir -p $O
MC_REQUIRE_REF=1 timeout 1500 python -m pytest tests -m gpu -x -q "$@" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 $O/pytest_gpu.log | grep -v amdgpu.ids
