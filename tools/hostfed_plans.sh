for o in "|" "seq_chunk=100|40,80,60,20" "seq_chunk=100|40,60,60,40" "seq_chunk=100|60,80,40,20" "seq_chunk=100|40,80,80" "seq_chunk=100|30,60,60,30,20" "seq_chunk=100|50,100,50" "seq_host_inorder=0|"; do
  opts="${o%%|*}"; plan="${o#*|}"
  if [ -n "$plan" ]; then export VSM_SEQ_PLAN="$plan"; else unset VSM_SEQ_PLAN; fi
  for mode in "pinned quiet resident_first" "quiet resident_first"; do
    echo -n "[$o] $mode: "
    VSM_PY_OPTIONS="$opts" timeout -k 10 120 python tools/hostfed_timeline.py $mode 2>&1 | grep TOOK | tail -4 | tr "\n" " "; echo
  done
done
