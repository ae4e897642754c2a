#!/bin/bash
# where the GPU hangs and what the CPU topology looks like on this box; the look-ahead call under CPU sets near / far from the GPU
lscpu | grep -i "numa\|socket\|model name" | head -12
for d in /sys/class/drm/card*/device; do [ -e $d/numa_node ] && echo "$d numa_node $(cat $d/numa_node) local_cpulist $(cat $d/local_cpulist) vendor $(cat $d/vendor)"; done
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | cut -c1-100
which numactl taskset
python3 - <<'PY'
import os
print("affinity now:", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], "...")
PY
BDF=$(python3 -c "
import torch
p = torch.cuda.get_device_properties(0)
print('%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))" 2>/dev/null)
echo "visible GPU at $BDF: numa_node $(cat /sys/bus/pci/devices/$BDF/numa_node) local_cpulist $(cat /sys/bus/pci/devices/$BDF/local_cpulist)"
if [ "$1" = "run" ]; then
  NEAR=$(cat /sys/bus/pci/devices/$BDF/local_cpulist)
  if [ "$(cat /sys/bus/pci/devices/$BDF/numa_node)" = "0" ]; then FAR=64-127,192-255; else FAR=0-63,128-191; fi
  for r in 1 2; do
    for set in none "$NEAR" "$FAR"; do
      if [ "$set" = none ]; then P=""; else P="taskset -c $set"; fi
      $P python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-per-frame --no-alone --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=sorted(d['step_ms_rank0'])
print('cpus %-22s value %6.0f  min %.2f p50 %.2f p90 %.2f max %.2f' % ('$set', d['value'], s[0], s[len(s)//2], s[int(len(s)*.9)], s[-1]))"
    done
  done
fi
