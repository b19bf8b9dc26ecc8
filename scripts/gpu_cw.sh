#!/bin/bash
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/cw; mkdir -p $O
timeout 400 python scripts/gpu_cbca_dense.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_instantiations.py -x -q -k "window or forms" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
