for combo in "0 2 4"; do set -- $combo
MC_CBCA_ABLATE=$1 MC_CBCA_PF=$2 MC_CBCA_RING=$3 timeout 600 python -m pytest tests -m gpu -q -x -k "cbca or golden or predict" 2>&1 | tail -1
MC_CBCA_ABLATE=$1 MC_CBCA_PF=$2 MC_CBCA_RING=$3 python bench.py --config mb_slow --steps 2 --warmup 1 --no-cpu-baseline --no-ref-gpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('ablate $1 PF $2 RING $3', j['ms_per_step'], j['stage_ms']['cbca'])"; 
MC_CBCA_ABLATE=$1 MC_CBCA_PF=$2 MC_CBCA_RING=$3 python bench.py --config kitti_slow --steps 3 --warmup 1 --no-cpu-baseline --no-ref-gpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('    kitti_slow', j['ms_per_step'], j['stage_ms']['cbca'])"
done
