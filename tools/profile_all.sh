# Regenerates the raw material of profiles/ on the GPU box (run through gpurun, from the repo root):
#   C=$(git rev-parse --short HEAD); gpurun --timeout 1100 -- "VSM_COMMIT=$C bash tools/profile_all.sh r05"
# Every artefact is stamped with the commit it was taken at.  Outputs: gpurun_out/prof/<tag>_* (copy into profiles/).
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SRC_SHA=$(cat $R/opencl-structure-from-motion_amd/csrc/*.hip $R/opencl-structure-from-motion_amd/csrc/*.h $R/opencl-structure-from-motion_amd/csrc/*.inc $R/opencl-structure-from-motion_amd/csrc/*.cpp | sha256sum | cut -c1-16)
STAMP="# commit: ${VSM_COMMIT:-unknown}"$'\n'"# sources: $SRC_SHA (sha256 of csrc/*.hip *.h *.inc *.cpp, first 16 hex digits)"$'\n'"# taken: $(date -u +%Y-%m-%dT%H:%MZ) on $(rocminfo 2>/dev/null | grep -m1 'Marketing Name' | sed 's/.*: *//') / $(grep -m1 'model name' /proc/cpuinfo | sed 's/.*: *//'), $(nproc) CPUs visible"
B="python3 $R/bench.py --no-cpu-baseline --no-verify --no-per-frame --no-alone"
stamp() { { echo "$STAMP"; echo "# command: $2"; cat "$1"; } > "$3"; }
# per-kernel time: the default form (GPU-resident) and the host-shared form (VSM_SEQ_V2=0)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 3 --warmup 1 > $O/stats.log 2>&1 &&
stamp $(ls $O/stats/*/*kernel_stats.csv) "rocprofv3 --kernel-trace --stats -- bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-per-frame" $O/${TAG}_lookahead_kernel_stats.csv &&
VSM_SEQ_V2=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -- $B --steps 3 --warmup 1 > $O/stats2.log 2>&1 &&
stamp $(ls $O/stats2/*/*kernel_stats.csv) "VSM_SEQ_V2=0 (host-shared form) rocprofv3 --kernel-trace --stats -- bench.py --steps 3 --warmup 1 ..." $O/${TAG}_lookahead_host_shared_kernel_stats.csv &&
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B --steps 1 --warmup 0 > $O/fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B --steps 1 --warmup 0 > $O/write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/sq -- $B --steps 1 --warmup 0 > $O/sq.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/tcc -- $B --steps 1 --warmup 0 > $O/tcc.log 2>&1 &&
cd $R && python tools/pmc_summary.py $(ls $O/fetch/*/*counter_collection.csv) $(ls $O/write/*/*counter_collection.csv) $O/pmc_hbm_raw.csv &&
stamp $O/pmc_hbm_raw.csv "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- bench.py --steps 1 --warmup 0 ...; traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB" $O/${TAG}_lookahead_pmc_hbm.csv &&
python tools/pmc_table.py $O/sq > $O/sq_table.txt && stamp $O/sq_table.txt "rocprofv3 --pmc SQ_* --kernel-trace -- bench.py --steps 1 --warmup 0 ..." $O/${TAG}_lookahead_pmc_sq.txt &&
python tools/pmc_table.py $O/tcc > $O/tcc_table.txt && stamp $O/tcc_table.txt "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -- bench.py ..." $O/${TAG}_lookahead_pmc_tcc.txt &&
cp $O/${TAG}_lookahead_pmc_hbm.csv profiles/ &&
timeout -k 10 500 python bench.py > $O/${TAG}_lookahead_bench.json 2> $O/bench.err &&
for t in 8 4 2 1; do VSM_HOST_THREADS=$t timeout -k 10 300 python bench.py --no-cpu-baseline --no-per-frame > $O/${TAG}_lookahead_bench_${t}threads.json 2>> $O/bench.err; done
VSM_SEQ_V2=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-per-frame > $O/${TAG}_lookahead_bench_host_shared_form.json 2>> $O/bench.err
echo "exit $?"; ls $O | grep ${TAG}_
