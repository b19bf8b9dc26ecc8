#!/bin/bash
# LDS latency / stalls and lane activity of the cbca tile kernel: bash scripts/gpu_pmc_tile2.sh <case>
ulimit -c 0
CASE=${1:-14natural}
O=$GRAFT_REPO_ROOT/gpurun_out/pmct2_$CASE; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES" "SQ_LEVEL_WAVES SQ_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $GRAFT_REPO_ROOT/scripts/gpu_cbca_tile.py $CASE --only-tile=0 --once > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $GRAFT_REPO_ROOT; python scripts/pmc_summary.py $O | grep -i "^Kernel\|tile_kernel" | cut -c1-1500 > $O/summary.csv; cat $O/summary.csv
