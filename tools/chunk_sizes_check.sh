for c in 7 20 33 100; do
  VSM_SEQ_CHUNK=$c timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-per-frame 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $c', d['value'], d['verified_bit_exact_vs_reference_hashes'])"
done
