#!/bin/bash
# Round latency at the reference's bench sizes: the C++ host mirror on the HIP engine (device link, both HBM layouts) beside the CPU port of
# the same two benches (oracle/bench_port.c) on this box's host, n = 10 ... 65536.  Output: one JSON line per run.
B=ark-mpc_amd/lib/arkmpc_host_bench
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
ARKMPC_MOCK_LINK=host $B mul_throughput 1000 3
ARKMPC_MOCK_LINK=host $B batch_ops 1000 10
