#!/bin/bash
# round 6, VERDICT item 8: the host-buffer path is 55 or 91 M docs/s depending on the session.  The box has two NUMA nodes; the GPU hangs on one.
set -u
tag=${1:-r06_numa}; O=$PWD/gpurun_out/$tag; mkdir -p $O
gn=$(rocm-smi --showtoponuma 2>/dev/null | grep "Numa Node:" | head -1 | sed "s/.*Numa Node: *//"); on=$((1 - gn))
lc=$(cat /sys/devices/system/node/node$gn/cpulist); fc=$(cat /sys/devices/system/node/node$on/cpulist)
{ echo "GPU on NUMA node $gn (cpus $lc); the other node: $on (cpus $fc)"
  echo "-- as the process comes"; timeout 300 python tools/host_numa_probe.py 2>&1 | tail -1
  echo "-- taskset to the GPU's node"; timeout 300 taskset -c $lc python tools/host_numa_probe.py 2>&1 | tail -1
  echo "-- taskset to the other node"; timeout 300 taskset -c $fc python tools/host_numa_probe.py 2>&1 | tail -1
  echo "-- as the process comes, again"; timeout 300 python tools/host_numa_probe.py 2>&1 | tail -1; } | tee $O/numa.txt
