#!/bin/bash
# hipGraph replay of a frame-step vs launch by launch (RVDD_GRAPH=0), same box, back to back.
# usage (GPU box, repo root): bash tools/graph_ab.sh > gpurun_out/graph_ab.log
for cfg in "C1 1" "C1 4" "C2 1" "C2 4"; do
  set -- $cfg
  for g in 0 1 0 1; do
    RVDD_GRAPH=$g timeout -k 10 200 python bench.py --config $1 --batch $2 --steps 10 --warmup 3 --cpu-frames 0 --no-kernel-events 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 B=$2 graphs=$g', d['value'], 'frames/s', d['ms_per_frame'], 'ms/frame')
"
  done
done
