# distribution of the step times of the look-ahead call over many steps (how often does a call take much longer than the median?)
run() { echo -n "$1: "; timeout -k 5 200 python bench.py --no-cpu-baseline --no-per-frame --no-alone --no-verify --steps ${STEPS:-200} --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms_rank0']); n=len(s); print(d['value'], 'median %.2f p90 %.2f p99 %.2f max %.2f; over 9 ms: %d of %d' % (s[n//2], s[int(n*0.9)], s[int(n*0.99)], s[-1], sum(1 for x in s if x > 9), n))"; }
run "default" 
VSM_HOST_THREADS=8 run "8 threads"
VSM_HOST_THREADS=4 run "4 threads"
