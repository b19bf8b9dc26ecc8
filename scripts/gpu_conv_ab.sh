#!/bin/bash
# several builds of libmcadcensus.so on ONE box: gpurun_in/lib<X>.so for X in $LIBS; mc_conv3x3 on the layer shapes, twice each
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
for L in $LIBS $LIBS; do
  cp gpurun_in/lib$L.so mc-cnn_amd/libmcadcensus.so
  echo "== lib$L"; timeout 300 python scripts/gpu_conv_bench.py --no-torch "$@" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_keep.so mc-cnn_amd/libmcadcensus.so
