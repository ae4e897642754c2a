#!/bin/bash
# HBM traffic per kernel of the look-ahead call (FETCH_SIZE and WRITE_SIZE in separate passes): tools/feat_pmc.sh TAG
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5/pmc_$TAG
rm -rf $O && mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-verify --no-per-frame --no-alone --startup 0"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B --steps 1 --warmup 0 "$@" > $O/fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B --steps 1 --warmup 0 "$@" > $O/write.log 2>&1 &&
cd $R && python tools/pmc_summary.py $(ls $O/fetch/*/*counter_collection.csv) $(ls $O/write/*/*counter_collection.csv) $O/pmc_hbm.csv && cat $O/pmc_hbm.csv | head -40
