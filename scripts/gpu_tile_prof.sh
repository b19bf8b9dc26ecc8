#!/bin/bash
# phase timeline of one block of the tile kernel: gpurun_in/lib<X>.so built with -DMC_TILE_PROF=<blockIdx>
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
for L in $LIBS; do cp gpurun_in/lib$L.so mc-cnn_amd/libmcadcensus.so; echo "== lib$L"; timeout 200 python scripts/gpu_tile_prof.py natural 2>&1 | grep -v amdgpu.ids; done
cp /tmp/lib_keep.so mc-cnn_amd/libmcadcensus.so
