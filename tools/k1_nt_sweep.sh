# K1 cache-policy sweep beside the current K2+K3 (ARKMPC_K1_NT bit0: x,y non-temporal; bit1: a,b; bit2: d||e stores)
for m in 1 0 2 3 5 1; do ARKMPC_K1_NT=$m python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-cold 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K1_NT=$m', 'ms_per_step', round(d['ms_per_step'],5), 'k1', round(d.get('pipeline',{}).get('k1_avg_launch_ms',0),5), 'k3', round(d['roofline']['avg_launch_ms'],5))"; done
