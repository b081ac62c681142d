#!/bin/bash
# The streaming host-to-host sessions by batch size, and the zero-copy phases against the copy pipeline on the same pinned vectors:
#     bash tools/e2e_by_batch_size.sh > profiles/<round>_e2e/e2e_by_batch_size.jsonl
R=$(cd "$(dirname "$0")/.." && pwd)
line() {   # $1 = log2n, $2 = ARKMPC_HOSTMUL_ZEROCOPY
  ARKMPC_HOSTMUL_ZEROCOPY=$2 python "$R/bench.py" --only-e2e --e2e-log2n $1 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); e = d.get('end_to_end', d)
r, p, t = e['one_party']['registered'], e['one_party']['pageable'], e['two_party_one_gpu']
print(json.dumps({'log2n': $1, 'zero_copy_phases': bool($2), 'registered_path': r.get('path'), 'registered_ms': r['ms'], 'registered_party_gates_per_s': r['party_gates_per_s'],
                  'registered_frac_of_measured_pcie': r['frac_of_measured_pcie'], 'pageable_new_vectors_ms': p['ms'], 'pageable_new_vectors_party_gates_per_s': p['party_gates_per_s'],
                  'two_party_one_gpu_ms': t['ms'], 'two_party_gates_per_s': t['two_party_gates_per_s'],
                  'two_threads_two_contexts_ms': t['two_host_threads_two_contexts']['ms'], 'measured_pcie_h2d_GBps': e['measured_pcie']['h2d_GBps'], 'check': e['results_check'][-2:]}))"
}
for L in 12 14 16 18 20 22; do line $L 1; done
for L in 16 20 22; do line $L 0; done
