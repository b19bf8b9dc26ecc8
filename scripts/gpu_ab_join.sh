#!/bin/bash
# StereoJoin: parity tests of the build in gpurun_in/libB.so, then the fast line of builds P and B on one box
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep2.so
cp gpurun_in/libB.so mc-cnn_amd/libmcadcensus.so
timeout 400 python -m pytest tests/test_ref_parity.py tests/test_gpu_parity.py -x -q -k "join or Join or predict or fast" 2>&1 | tail -2
cp /tmp/lib_keep2.so mc-cnn_amd/libmcadcensus.so
LIBS="P B" CFGS="kitti_fast" STEPS=30 bash scripts/gpu_ab_bench.sh
