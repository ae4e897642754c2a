run() { echo -n "$1: "; timeout -k 5 120 python bench.py --no-cpu-baseline --no-per-frame --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('verified_bit_exact_vs_reference_hashes'))"; }
for m in hhhhh hghhh hhghh hhhgh ghhhg hghgh gghhh hhhgg ggggg; do VSM_SEQ_V2=1 VSM_SEQ_FINAL_MASK=$m run "mask $m"; done
VSM_SEQ_V2=0 run v1
