for r in 1 2; do
for lib in default occ7; do
for c in 67 80 100; do
  if [ "$lib" = default ]; then unset VSM_LIB_PATH; else export VSM_LIB_PATH=$PWD/gpurun_variants/libvisomatch_$lib.so; fi
  VSM_SEQ_CHUNK=$c python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-per-frame --no-alone 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernel_avg_launch_us']
print('$lib', $c, d['value'], d['ms_per_step'], sorted(d['step_ms_rank0'])[:3], d['verified_bit_exact_vs_reference_hashes'], 'block', k.get('k_dc_block'), 'merge', k.get('k_dc_merge'))
"
done; done; done
