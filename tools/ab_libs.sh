#!/bin/bash
# interleaved bench runs of library variants: tools/ab_libs.sh "default w1 w2" [rounds]
ROUNDS=${2:-2}
for r in $(seq $ROUNDS); do
for v in $1; do
  if [ "$v" = default ]; then unset VSM_LIB_PATH; else export VSM_LIB_PATH=$PWD/gpurun_variants/libvisomatch_$v.so; fi
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-per-frame --no-alone 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernel_avg_launch_us']
print('$v', d['value'], d['ms_per_step'], sorted(d['step_ms_rank0'])[:3], d['verified_bit_exact_vs_reference_hashes'], 'block', k.get('k_dc_block'), 'merge', k.get('k_dc_merge'), 'match2', k.get('k_match<16>:pass2'), 'compact2', k.get('k_compact_matches:pass2'), 'refine', k.get('k_refine'))
"
done
done
