#!/bin/bash
# SQ counters of the bench kernels (MFMA utilisation, VALU activity, LDS bank conflicts), one --pmc pass each.
# usage (GPU box, repo root): bash tools/gpu_pmc_sq.sh <tag> [bench args...]
set -o pipefail
# single-GPU tool: `bench.py --gpus N` starts its ranks as child processes, and a launcher hop behind the profiler's
# preload (which has already initialised the GPU in the python process) is the re-exec this pool forbids
for a in "$@"; do case "$a" in --gpus|--gpus=*) echo "$0: do not pass --gpus (profile one rank: python3 bench.py ...)" >&2; exit 2;; esac; done
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-exact-ab --no-other-configs --frames 4 --no-kernel-events $*"
rc=0
# SQ_COUNTERS="A B ..." picks other counters (one the hardware does not know fails its own pass only)
for C in ${SQ_COUNTERS:-MfmaUtil SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU}; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- $BENCH > $OUT/$C.log 2>&1 || { rc=$?; echo "$C: pass failed ($rc)"; [ -n "$SQ_COUNTERS" ] || break; }
done
ls $OUT | head -20
exit $rc
