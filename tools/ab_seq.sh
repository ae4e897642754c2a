# Both look-ahead forms against the number of host threads on ONE box: frame-pairs/s, ms per 200-frame sequence, form, parity
#   /usr/local/graft/bin/gpurun -- 'bash tools/ab_seq.sh'
mkdir -p gpurun_out/r2
run() { echo -n "$1: "; timeout -k 5 120 python bench.py --no-cpu-baseline --no-per-frame --no-alone --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['lookahead_form'][:12], d.get('verified_bit_exact_vs_reference_hashes'), 'step spread %.1f %%' % (100 * (max(d['step_ms_rank0']) - min(d['step_ms_rank0'])) / min(d['step_ms_rank0'])))"; }
for t in 1 2 3 4 6 8 12 16; do VSM_HOST_THREADS=$t run "threads=$t GPU-resident (default)" || exit 1; done
for t in 2 4 8 12 16; do VSM_HOST_THREADS=$t VSM_SEQ_V2=0 run "threads=$t host-shared" || exit 1; done
