for g in 0 2 4 8; do VSM_MATCH_G=$g timeout -k 10 300 python bench.py --no-cpu-baseline --no-verify 2>/dev/null | python tools/kstat.py G$g >> gpurun_out/variants.log; done
