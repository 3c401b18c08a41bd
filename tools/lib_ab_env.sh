#!/bin/bash
# A/B of BUILDS of librvdd_hip.so and of environment switches on one box:
#   bash tools/lib_ab_env.sh "<bench args>" variant[:ENV=VAL[,ENV=VAL]] ...
# (variants are rvdd-release_amd/librvdd_hip_<variant>.so; the first one is restored at the end; three interleaved rounds;
#  the CPU-oracle samples and the other configurations of a default bench run are switched off: they are minutes per run)
ARGS="$1 --cpu-frames 0 --cpu-frames-8 0 --cpu-frames-wide 0 --no-exact-ab --no-other-configs"; shift
FIRST=${1%%:*}
cd rvdd-release_amd
for rep in 1 2 3; do for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  cp librvdd_hip_$v.so librvdd_hip.so
  (cd .. && env $envs timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('$spec', d['value'], 'frames/s ', ' '.join(f'{n.split(chr(60))[0]}={v_[\"avg_us\"]:.1f}' for n,v_ in k.items()))
")
done; done
cp librvdd_hip_$FIRST.so librvdd_hip.so
