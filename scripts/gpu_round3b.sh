#!/bin/bash
# one box: parity of the new tile kernel (with and without the plan), timing of several builds, fused-path checks
O=gpurun_out/r3b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cbca_tile.py -x -q > $O/pytest_tile.log 2>&1; tail -3 $O/pytest_tile.log
LIBS="${LIBS:-P B D}" bash scripts/gpu_abn.sh 14natural 5natural --only-tile=0 > $O/abn.log 2>&1; grep -v "^$" $O/abn.log | grep "==\|reads\|v0"
timeout 300 python scripts/gpu_fuzz.py ${FUZZ:-40} 43 > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
timeout 300 python bench.py --config mb_slow --pair natural --steps 3 --warmup 1 > $O/bench_mb_nat.json 2> $O/bench_mb_nat.err
timeout 300 python bench.py --config kitti_slow --steps 10 --warmup 2 > $O/bench_kitti_slow.json 2> $O/bench_kitti_slow.err
for c in mb_nat kitti_slow; do python -c "
import json; j=json.load(open('$O/bench_$c.json')); print('$c', j['value'], j['ms_per_step'], j['stage_ms'], j['roofline']['kernel'][:12], j['roofline']['frac'], j['verify']['bit_exact'], j['ops_ms_per_pair'])"; done
