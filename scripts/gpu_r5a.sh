#!/bin/bash
# Round 5, first GPU call (on the GPU box via gpurun): the whole -m gpu suite on the new routing, the two new microbenchmarks, and the A/B of
# the round's experiment builds (gpurun_in/lib<X>.so) on ONE box: P product, S8 / S4 SGM down sweep with 8 / 4 steps in flight, J2 / J1
# StereoJoin blocks of two / one waves, E tile kernel with the per-chunk event-row mask.  Output: gpurun_out/r5a/.
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r5a; mkdir -p $O
nproc > $O/nproc.txt
MC_REQUIRE_REF=1 timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 60 scripts/microbench/mfma_order.bin > $O/mfma_order.txt 2>&1; cat $O/mfma_order.txt
timeout 120 scripts/microbench/bw_sgm_layout.bin 370 1226 228 > $O/bw_sgm_layout_kitti.txt 2>&1; cat $O/bw_sgm_layout_kitti.txt
timeout 120 scripts/microbench/bw_sgm_layout.bin 1000 1500 256 > $O/bw_sgm_layout_mb.txt 2>&1; cat $O/bw_sgm_layout_mb.txt
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
use() { cp gpurun_in/lib$1.so mc-cnn_amd/libmcadcensus.so; }
# parity of the experiment builds on the tests that reach the changed kernel
for L in S8 S4; do use $L; MC_REQUIRE_REF=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sample_pair.py tests/test_gpu_fullsize.py -m gpu -x -q -k "sgm or predict or fast or fullsize or sample" > $O/pytest_$L.log 2>&1; echo "pytest($L) rc=$?"; tail -1 $O/pytest_$L.log; done
for L in J2 J1; do use $L; MC_REQUIRE_REF=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_ref_parity.py tests/test_gpu_sample_pair.py -m gpu -x -q -k "join or Join or fast or sample" > $O/pytest_$L.log 2>&1; echo "pytest($L) rc=$?"; tail -1 $O/pytest_$L.log; done
use E; MC_REQUIRE_REF=1 timeout 400 python -m pytest tests/test_gpu_cbca_tile.py tests/test_gpu_planned_routes.py tests/test_gpu_sample_pair.py -m gpu -x -q > $O/pytest_E.log 2>&1; echo "pytest(E) rc=$?"; tail -1 $O/pytest_E.log
line() { # lib config pair steps
  use $1
  timeout 300 python bench.py --config $2 ${3:+--pair $3} --steps $4 --warmup 2 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/ab_$1_$2_$3.json 2>/dev/null
  python -c "
import json; j=json.loads([l for l in open('$O/ab_$1_$2_$3.json') if l.startswith('{')][-1]); print('lib$1', '$2', '$3', j['ms_per_step'], j['ms_per_step_min'], {k: round(v, 3) for k, v in j['stage_ms'].items()})"
}
for rep in 1 2; do
  for L in P S8 S4 J2 J1; do line $L kitti_fast "" 30; done
  for L in P S8 E; do line $L kitti_slow "" 20; done
  for L in P E S8; do line $L mb_slow natural 3; done
done 2>&1 | tee $O/ab.txt
cp /tmp/lib_keep.so mc-cnn_amd/libmcadcensus.so
timeout 200 python scripts/gpu_cbca_tile.py 5natural 2>&1 | grep -v amdgpu.ids | tee $O/cbca_tile_kitti.txt
timeout 300 python bench.py --config mb_slow --pair mixed --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_mb_slow_mixed.json 2> $O/bench_mb_slow_mixed.err
python -c "
import json; j=json.loads([l for l in open('$O/bench_mb_slow_mixed.json') if l.startswith('{')][-1]); print('mixed', j['ms_per_step'], j['stage_ms'], j['verify']['bit_exact'], j['roofline']['kernels']['cbca'])"
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; j=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1]); n=j['north_star']
print('default', j['ms_per_step'], j['stage_ms'], j['roofline']['frac'], j['verify']['bit_exact'])
print('kitti_accurate', j['kitti_accurate']['ms_per_pair'], j['kitti_accurate']['stage_ms'], j['kitti_accurate']['verify']['bit_exact'])
print('north_star', n['ms_per_pair'], n['stage_ms'], n['per_volume'], n['verify']['bit_exact'])
for k in ('realistic_pair', 'realistic_pair_sample', 'mixed_pair'): print(k, n[k]['ms_per_pair'], n[k]['stage_ms'], n[k]['cbca_ms_per_launch'], n[k]['cbca_additions_per_voxel'], n[k]['verify']['bit_exact'])
print('fc', j['kitti_slow_fc'])"
ls $O
