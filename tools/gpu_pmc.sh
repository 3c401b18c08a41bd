#!/bin/bash
# One rocprofv3 --pmc pass per GROUP of counters (groups separated by ':', counters inside a group by ',';
# a group must fit the block's slots: 8 SQ counters, two of TA / TCP / TCC, FETCH_SIZE alone, WRITE_SIZE alone --
# "Request exceeds the capabilities of the hardware" otherwise, after which rocprofv3 hangs until the timeout).
# usage (GPU box, repo root): bash tools/gpu_pmc.sh <tag> "<group>:<group>..." <bench args...>
set -o pipefail
# single-GPU tool: `bench.py --gpus N` starts its ranks as child processes, and a launcher hop behind the profiler's
# preload (which has already initialised the GPU in the python process) is the re-exec this pool forbids
for a in "$@"; do case "$a" in --gpus|--gpus=*) echo "$0: do not pass --gpus (profile one rank: python3 bench.py ...)" >&2; exit 2;; esac; done
TAG=$1; GROUPS_=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# PMC_CMD: another program to count (e.g. "python3 $ROOT/tools/flow_bench.py"); the program itself, no launcher in front
BENCH=${PMC_CMD:-"python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-exact-ab --no-other-configs --frames 4 --no-kernel-events $*"}
rc=0; i=0
IFS=':' read -ra GS <<< "$GROUPS_"
for G in "${GS[@]}"; do
  i=$((i+1))
  # (a program that crashes in its exit handlers AFTER the profiler wrote its tables still counts: the table decides)
  timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc ${G//,/ } --output-format csv -d $OUT/g$i -- $BENCH > $OUT/g$i.log 2>&1
  prc=$?
  # a non-zero status is forgiven only for the known crash in the exit handlers: the program printed its JSON line (it ran to
  # its end) and the counter table exists; anything else stops here with the profiler's status
  if [ $prc -ne 0 ]; then
    if grep -q '^{' $OUT/g$i.log && ls $OUT/g$i/*/*_counter_collection.csv > /dev/null 2>&1; then :; else rc=$prc; tail -5 $OUT/g$i.log; break; fi
  fi
done
ls $OUT | head -20
exit $rc
