#!/bin/bash
# interleaved bench runs under environment settings: tools/ab_env.sh "A=1 B=2" "A=0" ... (each argument one setting; "-" = none)
for r in 1 2; do
for v in "$@"; do
  ( [ "$v" != "-" ] && export $v; python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-per-frame --no-alone 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernel_avg_launch_us']
print('$v', d['value'], d['ms_per_step'], sorted(d['step_ms_rank0'])[:3], d['verified_bit_exact_vs_reference_hashes'], 'prep', k.get('k_dc_prepare_kd_order'), 'block', k.get('k_dc_block'), 'merge', k.get('k_dc_merge'), 'match2', k.get('k_match<16>:pass2'), 'refine', k.get('k_refine'), 'compact2', k.get('k_compact_matches:pass2'))
" )
done
done
