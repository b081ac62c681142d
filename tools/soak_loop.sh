#!/bin/bash
# tools/soak_loop.sh [runs] -- the group-session soak in fresh processes, one line per run (result + seconds); host facts first.
# Used to chase a box-dependent flake (one failure in ~90 runs, seen only on a box where the run took 10x longer than usual).
runs=${1:-8}
mkdir -p gpurun_out/soak
{
  echo "kernel: $(uname -r) | amdgpu=$(cat /sys/module/amdgpu/version 2>/dev/null) | nproc=$(nproc) | load=$(cut -d" " -f1-3 /proc/loadavg) | cgroup_mem_max=$(cat /sys/fs/cgroup/memory.max 2>/dev/null)"
  echo "host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2) | thp=$(cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null) | numa_balancing=$(cat /proc/sys/kernel/numa_balancing 2>/dev/null) | iommu_groups=$(ls /sys/kernel/iommu_groups 2>/dev/null | wc -l) | cmdline=$(cat /proc/cmdline 2>/dev/null | tr ' ' '\n' | grep -E 'iommu|hugepage' | tr '\n' ' ') | mem_free_kb=$(grep -m1 MemAvailable /proc/meminfo | awk '{print $2}')"
  for i in $(seq 1 "$runs"); do
    t0=$(date +%s%N)
    timeout 600 python -m pytest tests/test_gpu_group_stream.py -m gpu -x -q -k soak 2>&1 | grep -E "Failed:|passed|failed|Error" | head -4
    ms=$(( ($(date +%s%N) - t0) / 1000000 ))
    echo "run $i: $ms ms"
    last=$ms
  done
  # the one failure seen so far was on a box where a run took 10-17 s instead of 2.7: stay on such a box and keep going
  if [ "${last:-0}" -gt 6000 ]; then
    echo "slow box: 40 more runs"
    for i in $(seq 1 40); do
      timeout 600 python -m pytest tests/test_gpu_group_stream.py -m gpu -x -q -k soak 2>&1 | grep -E "Failed:|passed|failed|Error" | head -4
    done
  fi
} > gpurun_out/soak/soak_$(date +%s).log 2>&1
cat gpurun_out/soak/soak_*.log | tail -40
