set -x
mkdir -p gpurun_out/r1a
O=$GRAFT_REPO_ROOT/gpurun_out/r1a
nproc > $O/nproc.txt; rocm-smi --showproductname > $O/smi.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 3 > $O/bench_kitti_fast.json 2> $O/bench_kitti_fast.err
timeout 300 python bench.py --config kitti_slow --steps 10 --warmup 2 > $O/bench_kitti_slow.json 2> $O/bench_kitti_slow.err
timeout 400 python bench.py --config mb_slow --steps 3 --warmup 1 > $O/bench_mb_slow.json 2> $O/bench_mb_slow.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_kitti_fast -o kitti_fast -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_kitti_fast.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_mb_slow -o mb_slow -- python $GRAFT_REPO_ROOT/bench.py --config mb_slow --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_mb_slow.log 2>&1
ls -R $O | head -50
