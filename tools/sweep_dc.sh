run() { timeout -k 10 200 python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for l in 32 80 100 128 160 200 256 400; do VSM_DC_LEAF=$l run "leaf$l"; VSM_DC_LEAF=$l run "leaf$l"; done
