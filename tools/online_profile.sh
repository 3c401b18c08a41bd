#!/bin/bash
# GPU box: kernel-trace statistics of the online-flow mode -> gpurun_out/online_<tag>_kernel_stats.csv
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/online_prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --online-flow --steps 1 --warmup 1 --cpu-frames 0 --no-exact-ab --no-other-configs > $OUT/run.log 2>&1
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
cp $f $ROOT/gpurun_out/online_${TAG}_kernel_stats.csv
grep -h '^{' $OUT/run.log | tail -1 | cut -c1-140
head -25 $ROOT/gpurun_out/online_${TAG}_kernel_stats.csv | cut -c1-150
