#!/bin/bash
# Every named configuration of bench.py on one GPU (C5: its per-GPU share), JSON lines under gpurun_out/<tag>_<cfg>.json.log
# usage (GPU box, repo root): bash tools/bench_all.sh <tag>
TAG=${1:-r02}
for cfg in C1 C2 C3 C4; do
  timeout -k 10 400 python bench.py --config $cfg --no-other-configs --steps 5 --warmup 2 --cpu-frames $([ $cfg = C4 ] && echo 4 || echo 10) 2>/dev/null | grep '^{' > gpurun_out/${TAG}_$cfg.json.log
  python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_$cfg.json.log').read())
print('$cfg', d['value'], 'frames/s  roofline', d['roofline']['frac'] if d['roofline'] else None, ' cpu', d['cpu_baseline']['value'] if d['cpu_baseline'] else None, 'psnr', d['task_psnr_db'], 'parity', d['cpu_baseline']['gpu_vs_cpu_parity_psnr_db'] if d['cpu_baseline'] else None)"
done
timeout -k 10 400 python bench.py --config C2 --batch 8 --steps 3 --warmup 1 --cpu-frames 0 2>/dev/null | grep '^{' > gpurun_out/${TAG}_C2_b8.json.log
timeout -k 10 400 python bench.py --config C5 --steps 2 --warmup 1 --cpu-frames 0 --collate-outputs 2>/dev/null | grep '^{' > gpurun_out/${TAG}_C5_share.json.log
for f in C2_b8 C5_share; do python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_$f.json.log').read()); print('$f', d['value'], d['config']['workload'], d['roofline']['frac'], d.get('collate'))"; done
