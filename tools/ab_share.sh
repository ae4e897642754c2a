# the host's share of the final stages inside the GPU-resident form: percent of every chunk's pairs x host threads, one box
mkdir -p gpurun_out/r2
run() { echo -n "$1: "; timeout -k 5 120 python bench.py --no-cpu-baseline --no-per-frame --no-alone --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['lookahead_form'][:12], d.get('verified_bit_exact_vs_reference_hashes'))"; }
for t in ${THREADS:-16 8}; do
  for s in ${SHARES:-0 10 20 30 40 50}; do VSM_HOST_THREADS=$t VSM_SEQ_HOST_SHARE=$s run "threads $t host share $s %" || exit 1; done
done
