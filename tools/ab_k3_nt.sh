# A/B of the K2+K3 store variants (ARKMPC_K3_NT: 1 = body's own non-temporal stores, 3 = LDS-staged whole-line stores behind a workgroup barrier,
# 4 = the same staged per wave, no barrier), interleaved to average out drift
for m in 3 4 3 4 1 3 4; do ARKMPC_K3_NT=$m python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-cold 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K3_NT=$m', 'ms_per_step', round(d['ms_per_step'],5), 'k1', round(d.get('pipeline',{}).get('k1_avg_launch_ms',0),5), 'k3', round(d['roofline']['avg_launch_ms'],5), 'ok', d.get('results_check', d.get('config',{}).get('results_check','?'))[:40])"; done
