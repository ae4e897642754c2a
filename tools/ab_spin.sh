# pool spin time x host threads: mean and spread of the step times (GPU-resident form)
run() { echo -n "$1: "; timeout -k 5 120 python bench.py --no-cpu-baseline --no-per-frame --no-alone --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms_rank0']); print(d['value'], d['ms_per_step'], 'min %.2f median %.2f max %.2f' % (s[0], s[len(s)//2], s[-1]))"; }
for t in 16 8; do for sp in 100 20 0 500; do VSM_HOST_THREADS=$t VSM_POOL_SPIN_US=$sp run "threads $t spin $sp us" || exit 1; done; done
