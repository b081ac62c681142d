for NT in 0 1 2 3 5 7; do
  ARKMPC_K1_NT=$NT python bench.py --steps 40 --warmup 5 --no-cpu-baseline --layout split 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K1_NT=$NT', '%.3e' % d['value'], 'k1 %.4f k3 %.4f dev %.4f pipefrac %.3f' % (d['pipeline']['k1_avg_launch_ms'], d['pipeline']['k3_avg_launch_ms'], d['pipeline']['device_ms_per_step'], d['pipeline']['frac_of_hbm_peak']))"
done
