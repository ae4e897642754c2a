#!/bin/bash
# the timed steps of bench.py one by one, a few processes in a row: tools/step_times.sh [runs] [options] [steps]
for r in $(seq ${1:-3}); do
  VSM_PY_OPTIONS="$2" timeout -k 10 200 python bench.py --no-cpu-baseline --no-per-frame --no-verify --no-alone --steps ${3:-30} --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], ' '.join('%.2f'%x for x in d['step_ms_rank0']))"
done
