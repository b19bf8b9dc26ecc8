#!/bin/bash
# Round 5: the horizontal SGM sweeps with the stores of SB steps / the refills of LB slots issued together (contiguous pieces of SB x ds floats), per launch, ONE box.
# Parity of two variants first.  Output: gpurun_out/r5p/.
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r5p; mkdir -p $O
cp mc-cnn_amd/libmcadcensus.so /tmp/lib_keep.so
use() { cp gpurun_in/lib$1.so mc-cnn_amd/libmcadcensus.so; }
for L in P SB4L4 SB8L8; do use $L; MC_REQUIRE_REF=1 timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sample_pair.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -m gpu -x -q -k "sgm or predict or fast or sample or golden" > $O/pytest_$L.log 2>&1; echo "pytest($L) rc=$?"; tail -1 $O/pytest_$L.log; done
cd /tmp && export TMPDIR=/tmp
for L in P SB2 SB4 SB8 L4 SB4L4 SB8L8 P; do
  cp $GRAFT_REPO_ROOT/gpurun_in/lib$L.so $GRAFT_REPO_ROOT/mc-cnn_amd/libmcadcensus.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$L -o p -- python $GRAFT_REPO_ROOT/bench.py --config kitti_fast --steps 10 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/prof_$L.log 2>&1
  python3 -c "
import csv
t=[(r['Name'],float(r['AverageNs'])/1e3) for r in csv.DictReader(open('$O/prof_$L/p_kernel_stats.csv')) if 'sgm_pass' in r['Name']]
t.sort(key=lambda x: x[0]); print('$L', '  '.join('%s %.1f' % (n.split('<')[1][:1], v) for n, v in t))"
done 2>&1 | tee $O/ab.txt
for L in P SB4L4 SB8L8; do
  cp $GRAFT_REPO_ROOT/gpurun_in/lib$L.so $GRAFT_REPO_ROOT/mc-cnn_amd/libmcadcensus.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profmb_$L -o p -- python $GRAFT_REPO_ROOT/bench.py --config mb_slow --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-ref-gpu --no-north-star --no-ops > $O/profmb_$L.log 2>&1
  python3 -c "
import csv
t=[(r['Name'],float(r['AverageNs'])/1e3) for r in csv.DictReader(open('$O/profmb_$L/p_kernel_stats.csv')) if 'sgm_pass' in r['Name']]
t.sort(key=lambda x: x[0]); print('mb $L', '  '.join('%s %.1f' % (n.split('<')[1][:1], v) for n, v in t))"
done 2>&1 | tee -a $O/ab.txt
cp /tmp/lib_keep.so $GRAFT_REPO_ROOT/mc-cnn_amd/libmcadcensus.so
