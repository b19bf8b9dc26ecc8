#!/bin/bash
# Several builds of libmcadcensus.so on ONE box (boxes of the pool differ by up to 25 %): gpurun_in/lib<X>.so for X in $LIBS
# usage: LIBS="A B C" bash scripts/gpu_abn.sh <args of scripts/gpu_cbca_tile.py>
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
for L in $LIBS $LIBS; do
  cp gpurun_in/lib$L.so mc-cnn_amd/libmcadcensus.so
  echo "== lib$L"; timeout 200 python scripts/gpu_cbca_tile.py "$@" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_keep.so mc-cnn_amd/libmcadcensus.so
