# A/B harness for the look-ahead path: interleaved runs in ONE box (box-to-box variance is larger than most effects),
# wall time per 200-frame sequence and CPU seconds per thread (tools/thread_cpu.py).  Edit the variants below.
#   /usr/local/graft/bin/gpurun -- 'bash tools/sweep_dc.sh'
run() { echo -n "$1: "; timeout -k 10 200 python tools/thread_cpu.py 150 2>/dev/null | grep -h "wall\|total" | tr '\n' ' '; echo; }
for i in 1 2 3; do
  run "default (final stage shared with the GPU)"
  VSM_DC_FULL=1 run "everything after the sort on the GPU"
  VSM_DC_GPU=0 run "final stage on the host pool only"
done
