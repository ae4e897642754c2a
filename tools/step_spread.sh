#!/bin/bash
# spread of the look-ahead call's time over many calls under environment settings: tools/step_spread.sh "A=1" "-" ...
for r in 1 2; do
for v in "$@"; do
  ( [ "$v" != "-" ] && export $v; python bench.py --steps ${STEPS:-40} --warmup 3 --no-cpu-baseline --no-per-frame --no-alone --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=sorted(d['step_ms_rank0'])
print('%-40s value %6.0f  min %.2f p25 %.2f p50 %.2f p75 %.2f p90 %.2f max %.2f (2nd %.2f; %d of %d over 1.5 x median) first steps %s' % ('$v', d['value'], s[0], s[len(s)//4], s[len(s)//2], s[3*len(s)//4], s[int(len(s)*.9)], s[-1], s[-2], sum(1 for x in s if x > 1.5 * s[len(s)//2]), len(s), d['step_ms_rank0'][:3]))
" )
done
done
