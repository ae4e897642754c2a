# chunk sizes for the GPU-resident look-ahead form on ONE box (interleaved; CHUNKS="67 80" REPS=3 STEPS=30 T=16)
run() { echo -n "$1: "; timeout -k 5 120 python bench.py --no-cpu-baseline --no-per-frame --no-alone --no-verify --steps ${STEPS:-12} --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms_rank0']); print(d['value'], d['ms_per_step'], 'min %.2f median %.2f max %.2f' % (s[0], s[len(s)//2], s[-1]))"; }
export VSM_HOST_THREADS=${T:-16}
for rep in $(seq 1 ${REPS:-2}); do for c in ${CHUNKS:-40 50 57 67 80 100}; do VSM_SEQ_CHUNK=$c run "chunk=$c" || exit 1; done; done
