#!/bin/bash
# A/B of SGM builds (gpurun_in/lib<X>.so for X in $LIBS) on one box: parity of the sweeps on each new build ($CHECK), then bench lines
O=gpurun_out/absgm; mkdir -p $O
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep0.so
for L in $CHECK; do
  cp gpurun_in/lib$L.so mc-cnn_amd/libmcadcensus.so
  MC_REQUIRE_REF=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_ref_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sample_pair.py -m gpu -x -q > $O/pytest_$L.log 2>&1; echo "pytest($L) rc=$?"; tail -2 $O/pytest_$L.log
done
cp /tmp/lib_keep0.so mc-cnn_amd/libmcadcensus.so
CFGS="${CFGS:-kitti_fast mb_slow}" STEPS=${STEPS:-10} bash scripts/gpu_ab_bench.sh
