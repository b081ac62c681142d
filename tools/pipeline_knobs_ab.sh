for a in "--chunks 1" "--chunks 2" "--k3-order 10" "--chunks 1"; do python bench.py $a --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-cold 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', 'value %.4e' % d['value'], 'ms_per_step', round(d['ms_per_step'],5), 'k1', round(d.get('pipeline',{}).get('k1_avg_launch_ms',0),5), 'k3', round(d['roofline']['avg_launch_ms'],5))"; done
for m in 0 2 1; do ARKMPC_K1_NT=$m python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-cold 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K1_NT=$m', 'value %.4e' % d['value'], 'ms_per_step', round(d['ms_per_step'],5), 'k1', round(d.get('pipeline',{}).get('k1_avg_launch_ms',0),5), 'k3', round(d['roofline']['avg_launch_ms'],5))"; done
