#!/bin/bash
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r2e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_instantiations.py -m gpu -q -x -k "lean" 2>&1 | tail -12
timeout 300 python scripts/gpu_cbca_bench.py 2>&1 | tee $O/cbca_bench.txt | grep -v amdgpu.ids
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "interp or mismatch or predict" 2>&1 | tail -3
