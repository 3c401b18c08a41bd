#!/bin/bash
# GPU box: TV-L1 tests, then the flow bench (stamps when the library was built with STAMPS=1) and the online-flow bench.
#   bash tools/tvl1_ab.sh <tag> [ENV=VAL ...]
TAG=${1:-x}; shift
for e in "$@"; do export "$e"; done
O=gpurun_out/tvl1_$TAG
timeout -k 10 300 python -m pytest tests/test_tvl1.py -m gpu -x -q > $O.tests.log 2>&1 || { tail -20 $O.tests.log; exit 1; }
tail -1 $O.tests.log
PAIRS=4 RVDD_TVL1_STAMPS=1 timeout -k 10 200 python tools/flow_bench.py > $O.flow.log 2>&1 || { tail -5 $O.flow.log; exit 1; }
grep stamps $O.flow.log | tail -6
tail -1 $O.flow.log | cut -c1-330
timeout -k 10 400 python bench.py --online-flow --steps 2 --warmup 1 --cpu-frames 0 --no-exact-ab --no-other-configs 2>/dev/null | tail -1 | cut -c1-140
