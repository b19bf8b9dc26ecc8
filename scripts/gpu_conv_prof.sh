#!/bin/bash
# usage: bash scripts/gpu_conv_prof.sh <tag> [shape index of scripts/gpu_conv_bench.py]
# rocprofv3 kernel trace + MFMA / stall counters of mc_conv3x3 alone (separate passes, kernel-trace only)
TAG=$1; IDX=${2:-1}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/gpu_conv_bench.py --no-torch --only $IDX --reps 10"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_conv$IDX -o conv$IDX -- $CMD > $O/prof_conv$IDX.log 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py $O/prof_conv$IDX/conv${IDX}_results.db > $O/conv${IDX}_kernel_stats.csv && cut -d, -f1-6,8-11 $O/conv${IDX}_kernel_stats.csv | head -6
cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc_conv$IDX/p$i -o p$i -- $CMD > $O/pmc_conv$IDX.p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $GRAFT_REPO_ROOT; python scripts/pmc_summary.py $O/pmc_conv$IDX > $O/conv${IDX}_pmc.csv; cat $O/conv${IDX}_pmc.csv
