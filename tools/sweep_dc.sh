# A/B runs of the look-ahead bench under environment switches, interleaved so that box-to-box variance cancels
run() { timeout -k 10 200 python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  for t in 12 14 15 16; do VSM_HOST_THREADS=$t run "threads$t"; done
done
grep -c throttled /sys/fs/cgroup/cpu.stat; cat /sys/fs/cgroup/cpu.stat
