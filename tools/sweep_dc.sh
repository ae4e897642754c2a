# A/B runs of the look-ahead path under environment switches, interleaved so that box-to-box variance cancels
run() { echo -n "$1: "; timeout -k 10 200 python tools/thread_cpu.py 150 2>/dev/null | grep -h "wall\|total" | tr '\n' ' '; echo; }
for i in 1 2 3; do
  run "current"
done
