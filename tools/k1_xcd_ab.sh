for m in 0 1 0 1; do ARKMPC_K1_XCD=$m python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-cold 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcd=$m', 'ms_per_step', round(d['ms_per_step'],5), 'k1', d.get('pipeline',{}).get('k1_avg_launch_ms'), 'k3', d['roofline']['avg_launch_ms'])"; done
