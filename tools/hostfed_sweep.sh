#!/bin/bash
# host-fed look-ahead call under option sets and chunk plans: tools/hostfed_sweep.sh [pinned] -- "opts|plan" ...
MODE=""
if [ "$1" = "pinned" ]; then MODE=pinned; shift; fi
[ "$1" = "--" ] && shift
for o in "$@"; do
  echo "== [$o] $MODE"
  opts="${o%%|*}"; plan=""
  [[ "$o" == *"|"* ]] && plan="${o#*|}"
  if [ -n "$plan" ]; then export VSM_SEQ_PLAN="$plan"; else unset VSM_SEQ_PLAN; fi
  VSM_PY_OPTIONS="$opts" timeout -k 10 120 python tools/hostfed_timeline.py $MODE 2>&1 | awk '/RUN 6/,0' | grep -v "^RUN"
done
