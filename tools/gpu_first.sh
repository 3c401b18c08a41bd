#!/bin/bash
# GPU pass: parity tests, smoke, a short bench, rocprof kernel stats
set -o pipefail
mkdir -p gpurun_out
export RVDD_TEST_NEXT=${RVDD_TEST_NEXT:-0}
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -30 \
&& timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tee gpurun_out/smoke.log | tail -5 \
&& timeout -k 10 600 python bench.py --steps 2 --warmup 1 --batch 1 --cpu-frames 2 2>&1 | tee gpurun_out/bench_first.log | tail -5 \
&& (cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_first -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 1 --cpu-frames 0 --frames 8 > $GRAFT_REPO_ROOT/gpurun_out/prof_first.log 2>&1; tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof_first.log)
