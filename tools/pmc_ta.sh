# texture-addresser / L1 counters of the look-ahead kernels (is a kernel bound by per-lane cache lookups?)
#   gpurun -- 'bash tools/pmc_ta.sh'      (two counters of one block per pass: more exceed what the hardware collects at once)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/ta
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-verify --no-per-frame --no-alone --steps 1 --warmup 0"
i=0
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout -k 5 60 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- $B > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log | cut -c1-300; exit 1; }
done
cd $R && for x in $O/p*/; do python tools/pmc_table.py $x; done > $O/table.txt; grep "k_match\|k_chain\|k_refine\|k_front\|k_nms\|k_filters\|k_emit" $O/table.txt
