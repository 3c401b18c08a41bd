#!/bin/bash
# GPU box: sample rocm-smi (power, clocks) twice a second while a bench config runs -> gpurun_out/power_<config>.log
CFG=${1:-C2}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
python bench.py --config $CFG --steps ${STEPS:-8} --warmup 1 --cpu-frames 0 --cpu-frames-8 0 --cpu-frames-wide 0 --no-exact-ab --no-other-configs > gpurun_out/power_${CFG}_bench.json 2>/dev/null &
BP=$!
for i in $(seq 1 60); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|junction" | tr '\n' ' '; echo
  kill -0 $BP 2>/dev/null || break
  sleep 0.5
done > gpurun_out/power_$CFG.log
wait $BP
tail -1 gpurun_out/power_${CFG}_bench.json | cut -c1-120
