#!/bin/bash
# A/B of several BUILDS of librvdd_hip.so on one box: bash tools/lib_ab.sh "<bench args>" variantA variantB ...
# (variants are rvdd-release_amd/librvdd_hip_<variant>.so; the first one is restored at the end)
ARGS=$1; shift
cd rvdd-release_amd
for rep in 1 2; do for v in "$@"; do cp librvdd_hip_$v.so librvdd_hip.so; (cd .. && timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('$v', d['value'], 'frames/s  frac', d['roofline']['frac'] if d['roofline'] else None, ' '.join(f'{n.split(chr(60))[0]}={v_[\"avg_us\"]:.1f}' for n,v_ in k.items()))
"); done; done
cp librvdd_hip_$1.so librvdd_hip.so
