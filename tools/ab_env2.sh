#!/bin/bash
# interleaved bench runs under environment settings, every main-stream kernel's average: tools/ab_env2.sh "A=1 B=2" "A=0" ... ("-" = none)
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do
for v in "$@"; do
  ( [ "$v" != "-" ] && export $v; python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-per-frame --no-alone 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernel_avg_launch_us']
short={'k_filters<false>':'filt','k_nms:dense':'nmsd','k_nms:sparse':'nmss','k_scan_cells':'scan','k_emit':'emit','k_bin_scan':'bscan','k_bin_scatter':'bscat','k_bin_rank':'brank','k_match<16>:pass1':'m1','k_compact_matches:pass1':'c1','k_match<16>:pass2':'m2','k_compact_matches:pass2':'c2','k_refine':'ref','k_front':'front','k_dc_keys':'keys','k_dc_vertex_sort':'ties','k_dc_prepare_kd_order':'prep','k_dc_block':'block','k_dc_merge':'merge','k_dc_support':'sup','k_dc_prior':'prior'}
print('$v'.replace('$PWD/gpurun_variants/libvisomatch_',''), '%.0f %.3f' % (d['value'], d['ms_per_step']), ['%.2f' % x for x in sorted(d['step_ms_rank0'])[:3]], 'OK' if d['verified_bit_exact_vs_reference_hashes'] else 'MISMATCH', ' '.join('%s=%.0f' % (short.get(a,a), b) for a,b in k.items()))
" )
done
done
