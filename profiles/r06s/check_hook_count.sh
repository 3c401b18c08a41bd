#!/bin/bash
# usage: dbg_count.sh variant [ENV=VAL ...]: events of the in-runtime check over 3 repetitions of 30 frames
v=$1; shift
cp rvdd-release_amd/librvdd_hip_$v.so rvdd-release_amd/librvdd_hip.so
env RVDD_GRAPH=0 RVDD_DEBUG_UPS=1 "$@" timeout -k 10 300 python tools/determinism_soak.py C2 ${DBG_REPS:-3} ${DBG_T:-30} > gpurun_out/dbg_$v.txt 2>&1
echo "== $v $*: $(grep -c 'UPSDBG launch' gpurun_out/dbg_$v.txt) events of $(grep -c . /dev/null) ; levels: $(grep 'UPSDBG launch' gpurun_out/dbg_$v.txt | awk '{print $5}' | sort | uniq -c | tr '\n' ' ') $(grep -m1 'poison launch' gpurun_out/dbg_$v.txt) $(grep -c fault gpurun_out/dbg_$v.txt) faults"
