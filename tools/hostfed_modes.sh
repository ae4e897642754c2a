#!/bin/bash
# the host-fed look-ahead call by process history (clean / after calls from HBM): tools/hostfed_modes.sh "opts|plan" ...
for o in "$@"; do
  opts="${o%%|*}"; plan=""
  [[ "$o" == *"|"* ]] && plan="${o#*|}"
  if [ -n "$plan" ]; then export VSM_SEQ_PLAN="$plan"; else unset VSM_SEQ_PLAN; fi
  for mode in "pinned quiet resident_first" "quiet resident_first" "pinned quiet" "quiet"; do
    echo -n "[$o] $mode: "
    VSM_PY_OPTIONS="$opts" timeout -k 10 120 python tools/hostfed_timeline.py $mode 2>&1 | grep TOOK | tail -4 | tr "\n" " "; echo
  done
done
