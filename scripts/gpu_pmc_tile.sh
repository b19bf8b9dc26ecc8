#!/bin/bash
# counters of the cbca tile kernel at 1000x1500x256 / 370x1226x228: bash scripts/gpu_pmc_tile.sh <case> [variant] [passes]
# (case = 14natural | 14smooth | 5natural | 5smooth, as scripts/gpu_cbca_tile.py names them; passes: how many of the counter sets)
ulimit -c 0
CASE=${1:-14natural}; VAR=${2:-0}; NP=${3:-3}
O=$GRAFT_REPO_ROOT/gpurun_out/pmct_$CASE; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  [ $i -gt $NP ] && break
  timeout 100 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $GRAFT_REPO_ROOT/scripts/gpu_cbca_tile.py $CASE --only-tile=$VAR --once > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $GRAFT_REPO_ROOT; python scripts/pmc_summary.py $O | grep -i "^Kernel\|tile_kernel" | cut -c1-1500 > $O/summary.csv; cat $O/summary.csv
