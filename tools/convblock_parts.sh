#!/bin/bash
# What a part of convblock_pipe_kernel costs in time, package power and clock: one library per switch set of
# tools/convblock_xp_patch.py (timing only: the results are wrong on purpose), C4 at B = 8 under rocm-smi sampling.
#   here (build):   bash tools/convblock_parts.sh build "0 1 2 4 8 6 15"
#   GPU box (run):  bash tools/convblock_parts.sh run   "0 1 2 4 8 6 15"  > gpurun_out/convblock_parts.txt
set -o pipefail
cd "$(dirname "$0")/.."
C=rvdd-release_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/scratch
  python3 tools/convblock_xp_patch.py tools/scratch/convnext_xp.hip > /dev/null || exit 1
  make -C $C > /dev/null || exit 1
  for xp in $2; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -DRVDD_XP=$xp -I$C -c tools/scratch/convnext_xp.hip -o tools/scratch/convnext_xp$xp.o || exit 1
    hipcc --offload-arch=gfx950 -shared -fPIC $C/conv3x3.o $C/conv3x3h.o $C/wino3x3.o tools/scratch/convnext_xp$xp.o $C/prestage.o $C/tvl1.o $C/srgb.o $C/runtime.o \
          -o rvdd-release_amd/librvdd_hip_xp$xp.so || exit 1
    echo built xp$xp
  done
  exit 0
fi
cp rvdd-release_amd/librvdd_hip.so /tmp/librvdd_hip_keep.so
for xp in $2; do
  cp rvdd-release_amd/librvdd_hip_xp$xp.so rvdd-release_amd/librvdd_hip.so
  python bench.py --config C4 --steps 8 --warmup 1 --cpu-frames 0 --cpu-frames-8 0 --cpu-frames-wide 0 --no-exact-ab --no-other-configs > /tmp/xp_bench.json 2>/dev/null &
  BP=$!
  : > /tmp/xp_power.log
  for i in $(seq 1 80); do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' ' >> /tmp/xp_power.log; echo >> /tmp/xp_power.log
    kill -0 $BP 2>/dev/null || break
    sleep 0.4
  done
  wait $BP
  python3 - $xp <<'PY'
import json, re, sys
xp = sys.argv[1]
d = [json.loads(l) for l in open('/tmp/xp_bench.json') if l.startswith('{')][-1]
k = d['kernels']; name = [n for n in k if n.startswith('convblock')][0]
rows = []
for l in open('/tmp/xp_power.log'):
    m = re.search(r'\((\d+)Mhz\).*Power \(W\): ([\d.]+)', l)
    if m: rows.append((int(m.group(1)), float(m.group(2))))
busy = sorted(rows, key=lambda r: -r[1])[:6]        # the six samples of highest power: inside the timed steps
mhz = sum(r[0] for r in busy) / max(len(busy), 1); w = sum(r[1] for r in busy) / max(len(busy), 1)
us = k[name]['avg_us']
print(f"xp {xp:>2}: {d['value']:7.1f} frames/s, block launch {us:7.1f} us = {us * mhz / 1e3:7.1f} k cycles, {mhz:5.0f} MHz, {w:6.0f} W")
PY
done
cp /tmp/librvdd_hip_keep.so rvdd-release_amd/librvdd_hip.so
