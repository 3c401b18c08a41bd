#!/bin/bash
# The two SQ counter groups used for the "where do a wave's cycles go" tables (wave cycles, MFMA-busy cycles, VALU /
# LDS / SALU instruction counts and active cycles, s_waitcnt cycles, LDS bank conflicts), summarised ON the box.
# usage (GPU box, repo root): bash tools/sq_pmc.sh <tag> <kernel-name-substring> <bench args...>
#   -> gpurun_out/<tag>_counters.json (+ one line per matching kernel on stdout); copy what is to be kept into profiles/
set -o pipefail
# single-GPU tool: `bench.py --gpus N` starts its ranks as child processes, and a launcher hop behind the profiler's
# preload (which has already initialised the GPU in the python process) is the re-exec this pool forbids
for a in "$@"; do case "$a" in --gpus|--gpus=*) echo "$0: do not pass --gpus (profile one rank: python3 bench.py ...)" >&2; exit 2;; esac; done
TAG=$1; KERN=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$PWD}
G="SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_VALU,SQ_BUSY_CYCLES,SQ_INSTS_VALU,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_WAVE_CYCLES:SQ_ACTIVE_INST_SCA,SQ_ACTIVE_INST_VMEM,SQ_INSTS_LDS,SQ_INSTS_MFMA,SQ_INSTS_SALU,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_VALU_MFMA_BUSY_CYCLES"
bash $ROOT/tools/gpu_pmc.sh $TAG "$G" "$@" || exit 1
cd $ROOT && python3 tools/pmc_any_summary.py $TAG $KERN && mv profiles/${TAG}_counters.json gpurun_out/ && rm -rf gpurun_out/pmc_$TAG
