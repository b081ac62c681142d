for lds in 0 26000 0 26000 0 26000 0 26000; do ARKMPC_TEST_DYN_LDS=$lds python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-cold 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dyn_lds=$lds', 'ms_per_step', round(d['ms_per_step'],5), 'k1', round(d.get('pipeline',{}).get('k1_avg_launch_ms',0),5), 'k3', round(d['roofline']['avg_launch_ms'],5))"; done
