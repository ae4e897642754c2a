# A/B runs of the look-ahead path under environment switches, interleaved so that box-to-box variance cancels
run() { echo -n "$1: "; timeout -k 10 200 python tools/thread_cpu.py 150 2>/dev/null | grep -h "wall\|total" | tr '\n' ' '; echo; }
for i in 1 2; do
  VSM_DC_LEAF=64 VSM_DC_TOP=0 run "leaf64top0"
  VSM_DC_LEAF=16 VSM_DC_TOP=120 run "leaf16top120"
  VSM_DC_LEAF=16 VSM_DC_TOP=240 run "leaf16top240"
  VSM_DC_LEAF=16 VSM_DC_TOP=480 run "leaf16top480"
  VSM_DC_LEAF=32 VSM_DC_TOP=240 run "leaf32top240"
done
VSM_DC_LEAF=16 VSM_DC_TOP=240 VSM_DEBUG_TIMING=1 timeout -k 10 200 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-per-frame 2> gpurun_out/dbg.log > /dev/null; grep "per pair\|seq:\|final stage" gpurun_out/dbg.log | sed -n 7,12p
