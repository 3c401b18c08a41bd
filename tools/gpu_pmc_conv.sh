#!/bin/bash
# SQ counters for the conv kernels (tools/conv_ab.py drives them back to back)
set -o pipefail
# single-GPU tool: `bench.py --gpus N` starts its ranks as child processes, and a launcher hop behind the profiler's
# preload (which has already initialised the GPU in the python process) is the re-exec this pool forbids
for a in "$@"; do case "$a" in --gpus|--gpus=*) echo "$0: do not pass --gpus (profile one rank: python3 bench.py ...)" >&2; exit 2;; esac; done
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/pmc_conv; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
VARIANTS=${VARIANTS:-0,3} timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -- python3 $ROOT/tools/conv_ab.py > $OUT/a.log 2>&1 \
&& VARIANTS=${VARIANTS:-0,3} timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU --output-format csv -d $OUT/b -- python3 $ROOT/tools/conv_ab.py > $OUT/b.log 2>&1
tail -3 $OUT/b.log
