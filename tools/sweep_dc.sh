# A/B runs of the look-ahead bench under environment switches, interleaved so that box-to-box variance cancels
run() { $PRE timeout -k 10 200 python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
lscpu | grep -i "numa\|model name\|socket\|l3"
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo
cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null
for i in 1 2; do
  PRE="" run "free"
  PRE="taskset -c 0-15" run "cpus0-15"
  PRE="taskset -c 0-31" run "cpus0-31"
  PRE="taskset -c 0-63" run "cpus0-63"
  PRE="taskset -c 64-127" run "cpus64-127"
done
