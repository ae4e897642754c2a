for sk in 0 1 2 3 4 8 16 32 63 0; do
  echo "== skip $sk"
  VSM_DC2_SKIP=$sk python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-per-frame --no-verify --no-alone 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], sorted(d['step_ms_rank0'])[:3])
"
done
