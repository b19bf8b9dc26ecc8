#!/bin/bash
# counters of the tile kernel's plan-writing and plan-reading passes: bash scripts/gpu_pmc_plan.sh <case> [passes]   (round 4: also used on the rolling form, history at b8c043d: profiles/r04_cbca_roll_pmc.txt)
ulimit -c 0
CASE=${1:-14natural}; NP=${2:-2}
O=$GRAFT_REPO_ROOT/gpurun_out/pmcp_$CASE; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LEVEL_WAVES SQ_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  [ $i -gt $NP ] && break
  timeout 100 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $GRAFT_REPO_ROOT/scripts/gpu_cbca_tile.py $CASE --plan-only --once > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $GRAFT_REPO_ROOT; python scripts/pmc_summary.py $O | grep -i "^Kernel\|tile_kernel\|roll_kernel" | cut -c1-1500 > $O/summary.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/summary.csv")))
for r in rows:
    name=r["Kernel"].replace("void mc::","")[:60]
    print(name)
    for k,v in r.items():
        if k not in ("Kernel",) and v not in ("nan",""):
            print("    %-28s %s" % (k, v))
PY
