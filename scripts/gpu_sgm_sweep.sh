#!/bin/bash
# SGM prefetch-depth sweep (tuning aid): prints the SGM stage time per (UH, UV) for two configs
for cfg in kitti_fast mb_slow; do
for UH in 4 8 16; do for UV in 4 8; do
  steps=20; [ $cfg = mb_slow ] && steps=2
  MC_SGM_UH=$UH MC_SGM_UV=$UV python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-ref-gpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$cfg UH=$UH UV=$UV ms/pair', j['ms_per_step'], 'sgm', j['stage_ms']['sgm'])"
done; done; done
