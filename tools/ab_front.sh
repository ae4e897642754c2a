for v in default fx1 fx2; do
  if [ "$v" = default ]; then unset VSM_LIB_PATH; else export VSM_LIB_PATH=$PWD/gpurun_variants/libvisomatch_$v.so; fi
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-per-frame --no-alone --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_avg_launch_us']
print('$v', d['value'], 'front', k.get('k_front'), 'dense', k.get('k_feat_dense'), 'refine', k.get('k_refine'))"
done
