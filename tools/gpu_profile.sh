#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the bench command, then HBM
# traffic counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage (on the GPU box, from the repo root): bash tools/gpu_profile.sh <tag> [bench args...]
set -o pipefail
# single-GPU tool: `bench.py --gpus N` starts its ranks as child processes, and a launcher hop behind the profiler's
# preload (which has already initialised the GPU in the python process) is the re-exec this pool forbids
for a in "$@"; do case "$a" in --gpus|--gpus=*) echo "$0: do not pass --gpus (profile one rank: python3 bench.py ...)" >&2; exit 2;; esac; done
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-exact-ab --no-other-configs $*"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1 \
&& timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH --frames 4 --no-kernel-events > $OUT/pmc_fetch.log 2>&1 \
&& timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH --frames 4 --no-kernel-events > $OUT/pmc_write.log 2>&1 \
&& timeout -k 10 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_clk -- $BENCH --frames 4 --no-kernel-events > $OUT/pmc_clk.log 2>&1
rc=$?
grep -h '^{' $OUT/stats.log | tail -1 | cut -c1-300
ls -R $OUT | head -40
exit $rc
