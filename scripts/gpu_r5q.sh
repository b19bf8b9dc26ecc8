#!/bin/bash
# Round 5, after the last product change (batched horizontal stores / refills): the -m gpu suite, the KITTI-fast evidence again (bench line, kernel stats, PMC traffic)
# and the driver's default line.  Output: gpurun_out/r5q/.
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r5q; mkdir -p $O
MC_REQUIRE_REF=1 timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 400 python bench.py --steps 30 --warmup 3 --no-north-star > $O/bench_kitti_fast.json 2> $O/bench_kitti_fast.err
timeout 400 python bench.py --config mb_slow --steps 5 --warmup 2 > $O/bench_mb_slow.json 2> $O/bench_mb_slow.err
python -c "
import json
for c in ('default','kitti_fast','mb_slow'):
    j=json.loads([l for l in open('$O/bench_%s.json' % c) if l.startswith('{')][-1]); print(c, j['value'], j['ms_per_step'], j['stage_ms'], j['roofline']['frac'], j['verify']['bit_exact'])"
bash scripts/gpu_prof.sh r5q kitti_fast 10 > /dev/null
bash scripts/gpu_prof.sh r5q mb_slow 2 > /dev/null
bash scripts/gpu_pmc.sh r5q kitti_fast 3 > /dev/null
timeout 600 python scripts/gpu_fuzz.py 150 777 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
