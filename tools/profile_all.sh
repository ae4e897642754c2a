# regenerates the raw material of profiles/ on the GPU box (run through gpurun); outputs under gpurun_out/prof/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-verify --no-per-frame"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 3 --warmup 1 > $O/stats.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B --steps 1 --warmup 0 > $O/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B --steps 1 --warmup 0 > $O/write.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/sq -- $B --steps 1 --warmup 0 > $O/sq.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/tcc -- $B --steps 1 --warmup 0 > $O/tcc.log 2>&1 &&
cd $R && python tools/pmc_summary.py $(ls $O/fetch/*/*counter_collection.csv) $(ls $O/write/*/*counter_collection.csv) $O/pmc_hbm.csv &&
python tools/pmc_table.py $O/sq > $O/sq_table.txt && python tools/pmc_table.py $O/tcc > $O/tcc_table.txt &&
cp $O/pmc_hbm.csv profiles/r01_lookahead_pmc_hbm.csv && python bench.py > $O/bench.json 2> $O/bench.err; echo "exit $?"; ls $O
