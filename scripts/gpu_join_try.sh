timeout 600 python -m pytest tests -m gpu -q -x -k "golden or predict or join or smoke" 2>&1 | tail -1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ref-gpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['stage_ms'])"
