#!/bin/bash
# usage (on the GPU box via gpurun): bash scripts/gpu_evidence.sh <tag>
# Everything a round's evidence needs: -m gpu tests, smoke, bench lines (default line with verify + north_star incl. its two
# realistic sub-records, the two accurate configurations, Middlebury size on the realistic pair, the accurate net from features),
# rocprofv3 kernel stats, PMC traffic / stall passes, the VALU issue-rate microbenchmark.
ulimit -c 0
TAG=${1:-ev}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
nproc > $O/nproc.txt
MC_REQUIRE_REF=1 timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 30 --warmup 3 > $O/bench_kitti_fast.json 2> $O/bench_kitti_fast.err
timeout 400 python bench.py --config kitti_slow --steps 20 --warmup 3 > $O/bench_kitti_slow.json 2> $O/bench_kitti_slow.err
timeout 400 python bench.py --config mb_slow --steps 5 --warmup 2 > $O/bench_mb_slow.json 2> $O/bench_mb_slow.err
timeout 400 python bench.py --config mb_slow --pair natural --steps 3 --warmup 1 > $O/bench_mb_slow_natural.json 2> $O/bench_mb_slow_natural.err
timeout 400 python bench.py --config kitti_slow_fc --steps 3 --warmup 1 > $O/bench_kitti_slow_fc.json 2> $O/bench_kitti_slow_fc.err
MC_BENCH_ONE_GPU=1 timeout 400 python bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks_one_gpu.err
timeout 400 python bench.py --config mb_slow --pair mixed --steps 3 --warmup 1 > $O/bench_mb_slow_mixed.json 2> $O/bench_mb_slow_mixed.err
for c in kitti_fast kitti_slow mb_slow mb_slow_natural mb_slow_mixed; do python -c "
import json; j=json.load(open('$O/bench_$c.json')); print('$c', j['value'], j['ms_per_step'], j['stage_ms'], j['roofline']['kernel'][:12], j['roofline']['frac'], j['verify']['bit_exact'], j['ops_ms_per_pair'], j['cpu_baseline']['value'] if j.get('cpu_baseline') else None)"; done
bash scripts/gpu_prof.sh $TAG kitti_fast 10 > /dev/null
bash scripts/gpu_prof.sh $TAG kitti_slow 5 > /dev/null
bash scripts/gpu_prof.sh $TAG mb_slow 2 > /dev/null
bash scripts/gpu_prof.sh $TAG mb_slow 1 natural > /dev/null
bash scripts/gpu_pmc.sh $TAG kitti_fast 3 > /dev/null
bash scripts/gpu_pmc.sh $TAG kitti_slow 2 > /dev/null
bash scripts/gpu_pmc.sh $TAG mb_slow 1 > /dev/null
bash scripts/gpu_pmc.sh $TAG mb_slow 1 natural > /dev/null
bash scripts/gpu_pmc.sh $TAG mb_slow 1 sample > /dev/null
bash scripts/gpu_pmc.sh $TAG mb_slow 1 mixed > /dev/null
timeout 900 python scripts/gpu_fuzz.py ${FUZZ_CASES:-200} ${FUZZ_SEED:-105} > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
[ -x scripts/microbench/valu_rate.bin ] && timeout 100 scripts/microbench/valu_rate.bin > $O/valu_rate.txt 2>&1
[ -x scripts/microbench/bw_sizes.bin ] && timeout 120 scripts/microbench/bw_sizes.bin > $O/bw_sizes.txt 2>&1
[ -x scripts/microbench/bw_lean.bin ] && timeout 120 scripts/microbench/bw_lean.bin > $O/bw_lean.txt 2>&1
ls $O
