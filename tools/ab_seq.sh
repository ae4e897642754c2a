# A/B of the two look-ahead forms on ONE box: value and step times at several host thread counts
#   /usr/local/graft/bin/gpurun -- 'bash tools/ab_seq.sh'
mkdir -p gpurun_out/r2
run() { echo -n "$1: "; python bench.py --no-cpu-baseline --no-per-frame --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms_rank0'], d.get('verified_bit_exact_vs_reference_hashes'))"; }
for t in 16 4 2; do
  VSM_HOST_THREADS=$t run "v2 threads=$t"
  VSM_HOST_THREADS=$t VSM_SEQ_V2=0 run "v1 threads=$t"
done
VSM_DEBUG_TIMING=1 python bench.py --no-cpu-baseline --no-per-frame --no-verify --steps 1 --warmup 1 2>&1 | grep -v "^{" | tail -30
