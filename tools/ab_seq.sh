# A/B of the look-ahead forms on ONE box: value and step times
#   /usr/local/graft/bin/gpurun -- 'bash tools/ab_seq.sh'
mkdir -p gpurun_out/r2
run() { echo -n "$1: "; timeout -k 5 120 python bench.py --no-cpu-baseline --no-per-frame --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['lookahead_form'][:12], d.get('verified_bit_exact_vs_reference_hashes'))"; }
for t in 1 2 3 4 8; do VSM_HOST_THREADS=$t run "auto threads=$t"; done
VSM_HOST_THREADS=2 VSM_SEQ_GPU_SORTS=0 run "threads=2, all sorts on the pool"
VSM_HOST_THREADS=2 VSM_SEQ_GPU_SORTS=100 run "threads=2, all sorts on the device"
VSM_HOST_THREADS=16 run "threads=16"
