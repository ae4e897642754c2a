# A/B runs of the look-ahead path under environment switches, interleaved so that box-to-box variance cancels
run() { echo -n "$1: "; timeout -k 10 200 python tools/thread_cpu.py 100 2>/dev/null | grep -h "wall\|total" | tr '\n' ' '; echo; }
for t in 16 8 4 2; do
  VSM_HOST_THREADS=$t VSM_DC_FULL=1 run "threads$t full"
  VSM_HOST_THREADS=$t VSM_DC_FULL=0 run "threads$t shared"
  VSM_HOST_THREADS=$t VSM_DC_GPU=0 run "threads$t host-only"
done
