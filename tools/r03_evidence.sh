#!/bin/bash
# Round-3 evidence in one GPU-box session (repo root): microbenchmarks behind the split-f16 kernels, every bench
# configuration, the A/B runs against the f32-MFMA kernels.  Outputs under gpurun_out/ev_*; copy what is kept to profiles/.
set -o pipefail
O=gpurun_out
mkdir -p $O
hipcc -O3 -Wno-unused-value --offload-arch=gfx950 tools/f16_mfma_bench.hip -o /tmp/f16b 2>/dev/null && timeout -k 10 120 /tmp/f16b > $O/ev_f16_mfma_bench.txt 2>&1
hipcc -O3 -Wno-unused-value --offload-arch=gfx950 tools/lds_b128_bench.hip -o /tmp/ldsb 2>/dev/null && timeout -k 10 120 /tmp/ldsb > $O/ev_lds_b128_bench.txt 2>&1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DRVDD_STAMPS -Wno-unused-value -Irvdd-release_amd/csrc tools/conv3x3h_bench.hip -o /tmp/c3hb 2>/dev/null && timeout -k 10 120 /tmp/c3hb > $O/ev_conv3x3h_stamps.txt 2>&1
B=4 VARIANTS=3,4 timeout -k 10 300 python tools/conv_ab.py > $O/ev_conv_ab_b4.txt 2>&1
timeout -k 10 300 python tools/online_flow_sensitivity.py 2>&1 | grep -E "vs|iterations|flow" > $O/ev_online_flow_sensitivity.txt
bash tools/bench_all.sh ev > $O/ev_bench_all.txt 2>&1
RVDD_CONV=f32 timeout -k 10 300 python bench.py --config C2 --cpu-frames 0 --cpu-frames-8 0 --cpu-frames-wide 0 2>/dev/null | grep '^{' > $O/ev_C2_f32kernels.json.log
RVDD_NEXT_SPLIT=0 timeout -k 10 300 python bench.py --config C4 --cpu-frames 0 --cpu-frames-8 0 --cpu-frames-wide 0 2>/dev/null | grep '^{' > $O/ev_C4_f32mlp.json.log
timeout -k 10 300 python bench.py --config C2 --online-flow --cpu-frames 0 --cpu-frames-8 0 --cpu-frames-wide 0 2>/dev/null | grep '^{' > $O/ev_C2_online_flow.json.log
cat $O/ev_bench_all.txt
python -c "
import json
for f in ('ev_C2_f32kernels','ev_C4_f32mlp','ev_C2_online_flow'):
    d=json.loads(open('$O/'+f+'.json.log').read()); print(f, d['value'], d['task_psnr_db'])"
