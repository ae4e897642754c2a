# A/B runs of the look-ahead path under environment switches, interleaved so that box-to-box variance cancels
run() { echo -n "$1: "; timeout -k 10 200 python tools/thread_cpu.py 150 2>/dev/null | grep -h "wall\|total" | tr '\n' ' '; echo; }
for i in 1 2; do
  VSM_SEQ_CHUNK=50 VSM_DC_TOP=240 run "chunk50 top240"
  VSM_SEQ_CHUNK=67 VSM_DC_TOP=240 run "chunk67 top240"
  VSM_SEQ_CHUNK=67 VSM_DC_TOP=120 run "chunk67 top120"
  VSM_SEQ_CHUNK=100 VSM_DC_TOP=240 run "chunk100 top240"
  VSM_SEQ_CHUNK=40 VSM_DC_TOP=240 run "chunk40 top240"
done
