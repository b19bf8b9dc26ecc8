#!/bin/bash
# round 6, step a: the resident-bank convolution -- parity, then the layer shapes against MIOpen
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r6a; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q > $O/pytest_conv.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_conv.log | grep -v amdgpu.ids
timeout 600 python scripts/gpu_conv_bench.py > $O/conv_bench.txt 2>&1; echo "bench rc=$?"
cat $O/conv_bench.txt | grep -v amdgpu.ids
