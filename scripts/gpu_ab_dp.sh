#!/bin/bash
# A/B of the pixel stride of the (H,W,ds) volumes (gpurun_in/libP.so: ds = D up to 4; libB.so: up to 32): parity subset on B, then bench lines
O=gpurun_out/abdp; mkdir -p $O
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep0.so
cp gpurun_in/libB.so mc-cnn_amd/libmcadcensus.so
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_sample_pair.py -m gpu -x -q > $O/pytest_B.log 2>&1; echo "pytest(B) rc=$?"; tail -3 $O/pytest_B.log
cp /tmp/lib_keep0.so mc-cnn_amd/libmcadcensus.so
LIBS="P B" CFGS="kitti_fast kitti_slow" STEPS=20 bash scripts/gpu_ab_bench.sh
