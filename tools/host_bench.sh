#!/bin/bash
# The reference's criterion bench definitions through the C++ host mirror on the HIP engine (host/bench_main.cpp), one JSON line per run.
#     tools/host_bench.sh            throughput sizes, the three link modes
#     tools/host_bench.sh latency    round latency at the reference's own sizes (device link, both HBM layouts) beside the CPU port of the
#                                    same benches (oracle/bench_port.c) on this box's host
B=ark-mpc_amd/lib/arkmpc_host_bench
if [ "${1:-}" = "latency" ]; then
  P=oracle/_build/bench_port
  export ARKMPC_MOCK_LINK=device
  for n in 10 100 1000 4096 16384 65536; do
    $B batch_ops $n 10
    ARKMPC_SHARE_LAYOUT=aos $B batch_ops $n 10 | sed 's/"link"/"layout": "aos", "link"/'
    $P batch_ops $n 10
  done
  for n in 100 1000 10000; do
    $B mul_throughput $n 3
    $P mul_throughput $n 3
  done
  exit 0
fi
for n in 10 100 1000 65536 1048576; do $B batch_ops $n 5; done
for n in 100 1000 10000; do $B mul_throughput $n 2; done
for n in 100 1000 10000; do $B msm_throughput $n 3; done
for link in device wire; do
  for n in 1000 1048576; do ARKMPC_MOCK_LINK=$link $B batch_ops $n 3; done
done
ARKMPC_MOCK_LINK=device $B mul_throughput 1000 2
ARKMPC_MOCK_LINK=device $B msm_throughput 10000 3
