#!/bin/bash
# headline of the look-ahead call under option sets: tools/sweep_opts.sh "seq_chunk=100" "seq_chunk=90,seq_first_chunk=60" ...
for o in "$@"; do
  v=$(VSM_PY_OPTIONS="$o" timeout -k 10 200 python bench.py --no-cpu-baseline --no-per-frame --no-verify --no-alone --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], min(d['step_ms_rank0']))")
  echo "$o -> $v"
done
