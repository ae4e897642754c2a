# A/B of chunk sizes for the GPU-resident look-ahead form on ONE box
#   /usr/local/graft/bin/gpurun -- 'bash tools/ab_chunk.sh'
mkdir -p gpurun_out/r2
run() { echo -n "$1: "; timeout -k 5 120 python bench.py --no-cpu-baseline --no-per-frame --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['lookahead_form'][:12], d.get('verified_bit_exact_vs_reference_hashes'))"; }
export VSM_SEQ_V2=1 VSM_HOST_THREADS=${T:-4}
for c in 25 50 67 100 200; do
  VSM_SEQ_CHUNK=$c run "chunk=$c taper" || exit 1
  VSM_SEQ_CHUNK=$c VSM_SEQ_TAPER=0 run "chunk=$c no taper" || exit 1
done
