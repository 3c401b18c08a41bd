#!/bin/bash
# GPU box: TV-L1 evidence -- the flow bench's JSON line (batches of eight 640x360 pairs), rocprofv3 kernel statistics of the
# same command, the online-flow bench line.  Summaries land in gpurun_out/<tag>_tvl1_* (copied to profiles/ afterwards).
TAG=${1:-r04f}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out
cd $ROOT
PAIRS=8 timeout -k 10 300 python tools/flow_bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_tvl1_flow_bench.json; cut -c1-400 gpurun_out/${TAG}_tvl1_flow_bench.json
timeout -k 10 400 python bench.py --online-flow --steps 3 --warmup 1 --cpu-frames 0 --no-exact-ab --no-other-configs 2>/dev/null | grep '^{' > gpurun_out/${TAG}_online_flow.json.log; cut -c1-140 gpurun_out/${TAG}_online_flow.json.log
OUT=$ROOT/gpurun_out/tvl1_prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PAIRS=8 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/flow_bench.py > $OUT/run.log 2>&1
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $ROOT/gpurun_out/${TAG}_tvl1_kernel_stats.csv && head -12 $ROOT/gpurun_out/${TAG}_tvl1_kernel_stats.csv | cut -c1-160
rm -rf $OUT/*/*trace.csv
