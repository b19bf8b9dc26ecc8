#!/bin/bash
# A/B of two builds of libmcadcensus.so on ONE box (boxes of the pool differ by up to 15 %): gpurun_in/libA.so, gpurun_in/libB.so
# usage: bash scripts/gpu_ab.sh <args of scripts/gpu_cbca_tile.py>
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
for L in A B A B; do
  cp gpurun_in/lib$L.so mc-cnn_amd/libmcadcensus.so
  echo "== lib$L"; timeout 200 python scripts/gpu_cbca_tile.py "$@" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_keep.so mc-cnn_amd/libmcadcensus.so
