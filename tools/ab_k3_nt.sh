for nt in 1 3 1 3 2 0; do
  echo "K3_NT=$nt"; ARKMPC_K3_NT=$nt python bench.py --no-extras --no-cpu-baseline --no-cold 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' value %.4e  k1 %.2f us  k3 %.2f us  frac %.3f  dev_ms %.4f  %s' % (d['value'], d['pipeline']['k1_avg_launch_ms']*1e3, d['pipeline']['k3_avg_launch_ms']*1e3, d['roofline']['frac'], d['pipeline']['device_ms_per_step'], d['results_check'][-4:]))"
done
