#!/bin/bash
# the bench's secondary legs under option sets: tools/legs.sh "" "fused_features=0,feat_order=0" ...
for o in "$@"; do
  VSM_PY_OPTIONS="$o" timeout -k 10 280 python bench.py --no-cpu-baseline --no-alone 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sc=d['secondary_configs']
print('[%s]' % '$o', 'headline', d['value'], '| per-frame', d['per_frame_api']['value'], '| vo', d['vo_process_api']['value'], '| host-in', d['lookahead_host_inputs']['value'],
      '| mono pf', sc['cfg3_640x480_mono_flow']['value'], 'la', sc['cfg3_640x480_mono_flow']['lookahead']['value'], '| cfg5', sc['cfg5_2048x1024_20k_dense']['lookahead']['value'], sc['cfg5_2048x1024_40k_dense']['lookahead']['value'],
      '| K8', d['vo_multi_sequence']['K8']['value'])"
done
