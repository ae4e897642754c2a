mkdir -p gpurun_out/r2
run() { echo -n "$1: "; timeout -k 5 120 python bench.py --no-cpu-baseline --no-per-frame --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['lookahead_form'][:12], d.get('verified_bit_exact_vs_reference_hashes'))"; }
export VSM_HOST_THREADS=${T:-16}
run "default (67,67,66)"
for l in "50,75,75" "40,80,80" "60,70,70" "70,70,60" "75,75,50" "80,70,50" "30,60,60,50" "80,80,40"; do VSM_SEQ_CHUNK=80 VSM_SEQ_CHUNKS=$l run "chunks $l" || exit 1; done
