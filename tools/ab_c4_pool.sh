#!/bin/bash
# A/B on one box: C4 with the pooling epilogue (default) and with the separate maxpool kernel, alternated.
set -e
mkdir -p gpurun_out
for i in 1 2; do
  for p in 1 0; do
    RVDD_NEXT_POOL=$p python bench.py --config C4 --cpu-frames 0 --cpu-frames-8 0 --cpu-frames-wide 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pool=$p', d['value'], d['roofline']['avg_launch_us'])" | tee -a gpurun_out/ab_pool.txt
  done
done
