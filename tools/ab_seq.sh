# A/B of the look-ahead forms on ONE box: value and step times
#   /usr/local/graft/bin/gpurun -- 'bash tools/ab_seq.sh'
mkdir -p gpurun_out/r2
run() { echo -n "$1: "; timeout -k 5 120 python bench.py --no-cpu-baseline --no-per-frame --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms_rank0'], d.get('verified_bit_exact_vs_reference_hashes'))"; }
for t in 16 8 4 2; do
VSM_SEQ_V2=1 VSM_HOST_THREADS=$t run "hybrid auto threads=$t"
done
VSM_SEQ_V2=1 VSM_SEQ_FINAL=host run "hybrid all-host threads=16"
VSM_SEQ_V2=1 VSM_SEQ_FINAL=gpu run "all-gpu threads=16"
VSM_SEQ_V2=0 run "v1 threads=16"
VSM_SEQ_V2=0 VSM_HOST_THREADS=8 run "v1 threads=8"
