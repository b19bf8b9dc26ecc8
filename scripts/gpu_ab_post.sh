#!/bin/bash
# A/B of post-processing builds (gpurun_in/lib<X>.so for X in $LIBS) on one box: parity of the disparity-image operators on each, then bench lines
O=gpurun_out/abpost; mkdir -p $O
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep0.so
for L in $LIBS; do
  cp gpurun_in/lib$L.so mc-cnn_amd/libmcadcensus.so
  MC_REQUIRE_REF=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_ref_parity.py -m gpu -x -q -k "post or mismatch or golden or ref" > $O/pytest_$L.log 2>&1; echo "pytest($L) rc=$?"; tail -1 $O/pytest_$L.log
done
cp /tmp/lib_keep0.so mc-cnn_amd/libmcadcensus.so
CFGS="${CFGS:-kitti_fast}" STEPS=20 bash scripts/gpu_ab_bench.sh
