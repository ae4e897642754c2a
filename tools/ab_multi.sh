#!/bin/bash
# the multi-sequence leg of bench.py under library variants: tools/ab_multi.sh "default prev" [rounds]
for r in $(seq ${2:-2}); do
for v in $1; do
  if [ "$v" = default ]; then unset VSM_LIB_PATH; else export VSM_LIB_PATH=$PWD/gpurun_variants/libvisomatch_$v.so; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alone 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', d['value'], {k:(v['value'], v['ms_per_step'], v['last_step_us']['pass1_us']) for k,v in d['vo_multi_sequence'].items() if isinstance(v,dict)})
"
done
done
