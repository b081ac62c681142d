# occupancy sensitivity of K1 / K2+K3: unused dynamic LDS caps the workgroups per CU (160 KB LDS per CU; K2+K3 holds 16 KB of its own):
# with the 16 KB kernel: 0 -> 4 waves per SIMD (the VGPR limit), 26000 -> 3, 50000 -> 2, 100000 -> 1; with the shipped 44 KiB allocation: 0 -> 3, 26000 -> 2, 50000+ -> 1
for lds in 0 26000 50000 100000 0; do ARKMPC_TEST_DYN_LDS=$lds python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-cold 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dyn_lds=$lds', 'ms_per_step', round(d['ms_per_step'],5), 'k1', round(d.get('pipeline',{}).get('k1_avg_launch_ms',0),5), 'k3', round(d['roofline']['avg_launch_ms'],5))"; done
