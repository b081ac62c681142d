#!/bin/bash
# The reference's criterion bench definitions through the C++ host mirror on the HIP engine (see host/bench_main.cpp).
B=ark-mpc_amd/lib/arkmpc_host_bench
for n in 10 100 1000 65536 1048576; do $B batch_ops $n 5; done
for n in 100 1000 10000; do $B mul_throughput $n 2; done
for n in 100 1000 10000; do $B msm_throughput $n 3; done
for link in device wire; do
  for n in 1000 1048576; do ARKMPC_MOCK_LINK=$link $B batch_ops $n 3; done
done
ARKMPC_MOCK_LINK=device $B mul_throughput 1000 2
ARKMPC_MOCK_LINK=device $B msm_throughput 10000 3
