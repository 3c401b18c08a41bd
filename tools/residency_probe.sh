#!/bin/bash
# per-dispatch wave residency of a kernel: SQ_WAVE_CYCLES (summed over waves) against GRBM_GUI_ACTIVE (the dispatch's
# GPU-busy cycles) and SQ_WAVES, one rocprofv3 --pmc pass; prints the largest dispatches of kernels matching $1
# usage (GPU box, repo root): bash tools/residency_probe.sh <kernel-substring> <bench args...>
set -o pipefail
# single-GPU tool: `bench.py --gpus N` starts its ranks as child processes, and a launcher hop behind the profiler's
# preload (which has already initialised the GPU in the python process) is the re-exec this pool forbids
for a in "$@"; do case "$a" in --gpus|--gpus=*) echo "$0: do not pass --gpus (profile one rank: python3 bench.py ...)" >&2; exit 2;; esac; done
KERN=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/resid
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-exact-ab --no-other-configs --frames 4 --no-kernel-events "$@" > $OUT/log 2>&1 || { tail -5 $OUT/log; exit 1; }
python3 - "$KERN" $OUT <<'PY'
import csv, glob, sys, collections
kern, out = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(dict)
for f in glob.glob(out + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            rows[(r["Dispatch_Id"], r["Kernel_Name"][:40])][r["Counter_Name"]] = float(r["Counter_Value"])
big = sorted(rows.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:8]
for (d, k), v in big:
    g, wc, w = v.get("GRBM_GUI_ACTIVE", 0), v.get("SQ_WAVE_CYCLES", 0), v.get("SQ_WAVES", 0)
    print(f"dispatch {d} {k}: GRBM_GUI_ACTIVE {g:.0f}  SQ_WAVES {w:.0f}  SQ_WAVE_CYCLES {wc:.0f}  wave-cycles x4 / waves / gui = {4*wc/max(w,1)/max(g,1):.3f}  SQ_BUSY_CYCLES {v.get('SQ_BUSY_CYCLES',0):.0f}")
PY
rm -rf $OUT
