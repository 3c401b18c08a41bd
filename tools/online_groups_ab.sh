#!/bin/bash
# GPU box: the online-flow mode of bench.py with the sequences as 1 / 2 groups (interleaved, same box) -> stdout
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
for rep in 1 2; do for g in 1 2; do
  timeout -k 10 300 python bench.py --online-flow --online-groups $g --steps ${STEPS:-3} --warmup 1 --cpu-frames 0 --no-exact-ab --no-other-configs 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('groups $g:', d['value'], 'frames/s', d['ms_per_step'], 'ms per step', d.get('gpu_clock_power'), 'psnr', d['task_psnr_db'])
"
done; done
