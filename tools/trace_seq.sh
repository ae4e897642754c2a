# kernel trace of the look-ahead call (timeline with tools/timeline.py); env passes through
#   gpurun -- 'VSM_SEQ_V2=1 bash tools/trace_seq.sh tagname'
TAG=${1:-trace}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/$TAG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --no-cpu-baseline --no-per-frame --no-verify --no-alone --steps 2 --warmup 2 > $O.log 2>&1
echo "exit $?"
