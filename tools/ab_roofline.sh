#!/bin/bash
# the bench line's headline and roofline under option sets (full bench: the work counters come from the cpu_baseline leg)
for o in "$@"; do
  VSM_PY_OPTIONS="$o" timeout -k 10 400 python bench.py --no-per-frame 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('[$o]', d['value'], d['ms_per_step'], 'frac', r['frac'], r['avg_launch_us'], 'alone', r['alone']['frac'], r['alone']['avg_launch_us'], 'all-profiled', r['with_every_kernel_profiled']['avg_launch_us'])"
done
