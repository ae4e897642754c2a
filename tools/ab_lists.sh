# explicit chunk lists for the GPU-resident form (200 frames), interleaved on one box
run() { echo -n "$1: "; timeout -k 5 120 python bench.py --no-cpu-baseline --no-per-frame --no-alone --no-verify --steps ${STEPS:-20} --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms_rank0']); print(d['value'], d['ms_per_step'], 'min %.2f median %.2f max %.2f' % (s[0], s[len(s)//2], s[-1]))"; }
export VSM_HOST_THREADS=${T:-16} VSM_SEQ_CHUNK=100
for rep in 1 2; do for l in ${LISTS:-76,76,48 80,72,48 72,80,48 84,76,40 76,84,40 70,70,60 90,70,40 66,66,68 60,70,70}; do VSM_SEQ_CHUNKS=$l run "chunks $l" || exit 1; done; done
