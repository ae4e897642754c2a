for c in 48 64 80 96 112; do VSM_SEQ_CHUNK=$c timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --no-per-frame 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chunk $c value', d['value'], 'verified', d['verified_bit_exact_vs_reference_hashes'], d['step_ms_rank0'], d['sequence_timings_us'])" >> gpurun_out/variants.log; done
