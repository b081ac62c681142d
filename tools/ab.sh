#!/bin/bash
# One parametrised A/B runner for the engine's environment knobs (DESIGN.md "Environment switches"), interleaved to average out drift:
#     tools/ab.sh VAR v1 v2 v3 ... [-- extra bench.py arguments]      e.g.  tools/ab.sh ARKMPC_K3_NT 3 4 3 4 1      (K2+K3 store variants)
#                                                                            tools/ab.sh ARKMPC_K1_LDS 0 40000 80000 160000   (K1 occupancy cap)
#                                                                            tools/ab.sh ARKMPC_TEST_DYN_LDS 0 26000 50000    (K2+K3 occupancy cap)
#                                                                            tools/ab.sh ARKMPC_K1_NT 1 0 2 3 5 -- --layout aos
#     tools/ab.sh --cmd VAR v1 v2 ... -- <command>     runs <command> under each value instead of bench.py (tools/ec_bench.py, tools/ed_bench.py,
#                                                       tools/msm_bench.py, tools/kernel_suite.py ...) and prints its last line
#     tools/ab.sh --trace <outdir> -- <command>        rocprofv3 --kernel-trace --stats of <command>, top kernels printed (per-kernel durations)
# It replaces the one-off *_ab.sh / *_probe.sh / *_profile.sh scripts of rounds 1-3 (their results are under profiles/r0*).
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "${1:-}" = "--trace" ]; then
  O=$2; shift 3
  case "$O" in /*) ;; *) O="$PWD/$O";; esac
  CMD=(); for a in "$@"; do case "$a" in tools/*|probes/*|bench.py) CMD+=("$R/$a");; *) CMD+=("$a");; esac; done; set -- "${CMD[@]}"
  export TMPDIR=/tmp; mkdir -p "$O"; (cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d "$O" -o p -- "$@" > "$O/cmd.out" 2> "$O/cmd.err")
  python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:16] if f else []:
    print("%-64s calls %6s avg_us %10.1f total_ms %9.3f %5s%%" % (r["Name"].split("(")[0][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
  exit 0
fi
CMD=0; if [ "${1:-}" = "--cmd" ]; then CMD=1; shift; fi
VAR=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ $# -gt 0 ] && shift
for v in "${VALS[@]}"; do
  if [ $CMD = 1 ]; then
    echo "$VAR=$v  $(env $VAR=$v "$@" 2>/dev/null | tail -1)"
  else
    env $VAR=$v python "$R/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-cold "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', 'value %.4e' % d['value'], 'ms_per_step %.5f' % d['ms_per_step'], 'k1 %.5f' % d['k1_avg_launch_ms'], 'k3 %.5f' % d['roofline']['avg_launch_ms'], 'frac %.4f' % d['roofline']['frac'], d['results_check'][-2:])"
  fi
done
