mkdir -p gpurun_out/r2
run() { echo -n "$1: "; timeout -k 5 120 python bench.py --no-cpu-baseline --no-per-frame --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['lookahead_form'][:12], d.get('verified_bit_exact_vs_reference_hashes'))"; }
for t in 10 12 14 16; do
  VSM_HOST_THREADS=$t VSM_SEQ_V2=0 run "threads $t v1" || exit 1
  VSM_HOST_THREADS=$t VSM_SEQ_V2=1 run "threads $t v2" || exit 1
done
