ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/pmcm; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout 60 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $GRAFT_REPO_ROOT/scripts/gpu_cbca_tile.py 14smooth --only-tile=0 --once > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $GRAFT_REPO_ROOT; python scripts/pmc_summary.py $O | grep -i "^Kernel\|tile_kernel" | cut -c1-900
