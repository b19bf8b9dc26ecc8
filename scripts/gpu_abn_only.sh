#!/bin/bash
# tile-kernel parity tests on the tree's build, then timing of several builds (gpurun_in/lib<X>.so) on the same box
O=gpurun_out/r3b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cbca_tile.py -x -q > $O/pytest_tile.log 2>&1; tail -2 $O/pytest_tile.log
LIBS="${LIBS:-P B}" bash scripts/gpu_abn.sh ${CASES:-14natural 5natural} --only-tile=0 > $O/abn2.log 2>&1; grep -v "^$" $O/abn2.log | grep "==\|reads\|v0"
