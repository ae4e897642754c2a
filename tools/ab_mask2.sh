# chunk size x per-chunk placement of the final stage (GPU-resident pipeline), 16 host threads
mkdir -p gpurun_out/r2
run() { echo -n "$1: "; timeout -k 5 120 python bench.py --no-cpu-baseline --no-per-frame --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['lookahead_form'][:12], d.get('verified_bit_exact_vs_reference_hashes'))"; }
export VSM_SEQ_V2=1 VSM_SEQ_TAPER=0 VSM_HOST_THREADS=${T:-16}
for cm in "25:dhdhdhdh" "25:hdhdhdhd" "25:ddhddhdd" "25:hhdhhdhh" "34:dhdhdh" "34:hdhdhd" "40:hdhdh" "40:dhdhd" "50:hdhd" "50:dhdh" "50:dddd" "25:dddddddd"; do
  c=${cm%%:*}; m=${cm#*:}
  VSM_SEQ_CHUNK=$c VSM_SEQ_FINAL_MASK=$m run "chunk=$c mask=$m" || exit 1
done
