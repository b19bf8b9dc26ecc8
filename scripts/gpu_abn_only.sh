#!/bin/bash
O=gpurun_out/r3b; mkdir -p $O
LIBS="${LIBS:-P B}" bash scripts/gpu_abn.sh 14natural 5natural --only-tile=0 > $O/abn2.log 2>&1; grep -v "^$" $O/abn2.log | grep "==\|reads\|v0"
