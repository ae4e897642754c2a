#!/bin/bash
# kernel times of the bench's timed calls under option sets: tools/ab_opts_kernels.sh "" "match_heads=0"
for o in "$@"; do
VSM_PY_OPTIONS="$o" timeout -k 10 280 python bench.py --no-cpu-baseline --no-per-frame 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_avg_launch_us']
rb=d.get('roofline_by_kernel') or {}
print('[$o]', d['value'], {n:k[n] for n in k if 'match' in n or 'feat' in n or 'front' in n or 'refine' in n}, 'alone', {n:rb[n].get('alone_us') for n in rb if 'match' in n})"
done
