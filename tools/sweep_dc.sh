# A/B runs of the look-ahead path under environment switches, interleaved so that box-to-box variance cancels
run() { echo -n "$1: "; timeout -k 10 200 python tools/thread_cpu.py 150 2>/dev/null | grep -h "wall\|total" | tr '\n' ' '; echo; }
for i in 1 2 3; do
  run "default"
done
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-per-frame 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'])"
