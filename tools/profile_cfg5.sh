# Per-kernel evidence for BASELINE.json configs[4] (2048x1024 stereo, 40 k dense features per image) through the look-ahead call:
#   C=$(git rev-parse --short HEAD); gpurun --timeout 900 -- "VSM_COMMIT=$C bash tools/profile_cfg5.sh r04"
# Outputs: gpurun_out/prof5/<tag>_cfg5_kernel_stats.csv, <tag>_cfg5_pmc_hbm.csv (copy into profiles/).
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof5
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=5
SRC_SHA=$(cat $R/opencl-structure-from-motion_amd/csrc/*.hip $R/opencl-structure-from-motion_amd/csrc/*.h $R/opencl-structure-from-motion_amd/csrc/*.inc $R/opencl-structure-from-motion_amd/csrc/*.cpp | sha256sum | cut -c1-16)
STAMP="# commit: ${VSM_COMMIT:-unknown}"$'\n'"# sources: $SRC_SHA (sha256 of csrc/*.hip *.h *.inc *.cpp, first 16 hex digits)"$'\n'"# taken: $(date -u +%Y-%m-%dT%H:%MZ) on $(rocminfo 2>/dev/null | grep -m1 'Marketing Name' | sed 's/.*: *//'), $(nproc) CPUs visible"
stamp() { { echo "$STAMP"; echo "# command: $2"; cat "$1"; } > "$3"; }
B="python3 $R/tools/cfg5_chunks.py cfg5_2048x1024_quad 120 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1 &&
stamp $(ls $O/stats/*/*kernel_stats.csv) "rocprofv3 --kernel-trace --stats -- tools/cfg5_chunks.py cfg5_2048x1024_quad 120 0 (120 frames 2048x1024, 4 look-ahead calls, the library's own chunking)" $O/${TAG}_cfg5_kernel_stats.csv &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B > $O/write.log 2>&1 &&
cd $R && python tools/pmc_summary.py $(ls $O/fetch/*/*counter_collection.csv) $(ls $O/write/*/*counter_collection.csv) $O/pmc_hbm_raw.csv &&
stamp $O/pmc_hbm_raw.csv "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- tools/cfg5_chunks.py cfg5_2048x1024_quad 120 0; traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB" $O/${TAG}_cfg5_pmc_hbm.csv
echo "exit $?"; cat $O/stats.log | grep cfg5; ls $O | grep ${TAG}_
