#!/bin/bash
# round 6, step f: microbenchmark of the horizontal SGM launch with writer waves
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r6f; mkdir -p $O
cd scripts/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 bw_sgm_writer.hip -o /tmp/bw_sgm_writer.bin || exit 1
timeout 300 /tmp/bw_sgm_writer.bin 370 1226 228 > $O/bw_sgm_writer_kitti.txt 2>&1; cat $O/bw_sgm_writer_kitti.txt
timeout 300 /tmp/bw_sgm_writer.bin 1000 1500 256 > $O/bw_sgm_writer_mb.txt 2>&1; cat $O/bw_sgm_writer_mb.txt
