#!/bin/bash
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r6b; mkdir -p $O; rm -f $O/abl.txt
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -2
for d in 1 0; do
echo "dbg=$d" >> $O/abl.txt
MC_CONV_DBG=$d timeout 600 python scripts/gpu_conv_bench.py --no-torch >> $O/abl.txt 2>&1
done
grep -v amdgpu.ids $O/abl.txt
