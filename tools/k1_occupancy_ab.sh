for lds in 0 -1 0 -1 0 -1; do if [ $lds = -1 ]; then unset ARKMPC_K1_LDS; else export ARKMPC_K1_LDS=$lds; fi; python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-cold 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k1_lds=$lds', 'value %.4e' % d['value'], 'ms_per_step', round(d['ms_per_step'],5), 'k1', round(d.get('pipeline',{}).get('k1_avg_launch_ms',0),5), 'k3', round(d['roofline']['avg_launch_ms'],5), 'frac', round(d['roofline']['frac'],4))"; done
python bench.py --layout aos --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-cold 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('aos default', 'value %.4e' % d['value'], 'k1', round(d.get('pipeline',{}).get('k1_avg_launch_ms',0),5), 'k3', round(d['roofline']['avg_launch_ms'],5))"
ARKMPC_K1_LDS=0 python bench.py --layout aos --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-cold 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('aos k1_lds=0', 'value %.4e' % d['value'], 'k1', round(d.get('pipeline',{}).get('k1_avg_launch_ms',0),5), 'k3', round(d['roofline']['avg_launch_ms'],5))"
