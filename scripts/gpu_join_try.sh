#!/bin/bash
# tuning aid: parity subset for the fused fast path + kitti_fast bench (MC_JOIN_KERNEL=1 = compute-once kernel;
# MC_JOIN_ABLATE bits on the owner-tile kernel: 1 = no line stores, 2 = no ring writes, 4 = no MFMAs)
timeout 600 python -m pytest tests -m gpu -q -x -k "golden or predict or join or smoke" 2>&1 | tail -1
for a in 0 7; do MC_JOIN_ABLATE=$a python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ref-gpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('ablate $a', j['ms_per_step'], j['stage_ms']['join'])"; done
