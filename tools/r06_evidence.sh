#!/bin/bash
# Round-6 evidence session on the GPU box: driver-form bench line, rocprofv3 kernel stats + PMC traffic + SQ counters of the
# same build for C2, kernel stats + traffic for C4, power / clock log.  Summaries land in gpurun_out/ (and are copied to profiles/).
TAG=${1:-r06z}
mkdir -p gpurun_out

echo "== C2 profile"; bash tools/gpu_profile.sh ${TAG} > gpurun_out/${TAG}_profile.log 2>&1; tail -2 gpurun_out/${TAG}_profile.log | cut -c1-200
python tools/pmc_summary.py ${TAG} C2 > /dev/null && cp profiles/${TAG}_pmc_traffic.json gpurun_out/${TAG}_c2_pmc_traffic.json
cp gpurun_out/prof_${TAG}/stats/*/*kernel_stats.csv gpurun_out/${TAG}_c2_kernel_stats.csv
rm -rf gpurun_out/prof_${TAG}/pmc_*/*/*.csv gpurun_out/prof_${TAG}/stats/*/*trace.csv
echo "== C2 SQ counters"; bash tools/gpu_pmc_sq.sh ${TAG} > gpurun_out/${TAG}_sq.log 2>&1; python tools/pmc_sq_summary.py ${TAG} > /dev/null && cp profiles/${TAG}_sq_counters.json gpurun_out/${TAG}_c2_sq_counters.json
rm -rf gpurun_out/sq_${TAG}/*/*/*.csv
echo "== C4 profile"; bash tools/gpu_profile.sh ${TAG}c4 --config C4 > gpurun_out/${TAG}c4_profile.log 2>&1; tail -2 gpurun_out/${TAG}c4_profile.log | cut -c1-200
python tools/pmc_summary.py ${TAG}c4 C4 > /dev/null && cp profiles/${TAG}c4_pmc_traffic.json gpurun_out/${TAG}_c4_pmc_traffic.json
cp gpurun_out/prof_${TAG}c4/stats/*/*kernel_stats.csv gpurun_out/${TAG}_c4_kernel_stats.csv
rm -rf gpurun_out/prof_${TAG}c4/pmc_*/*/*.csv gpurun_out/prof_${TAG}c4/stats/*/*trace.csv
cp profiles/traffic.json gpurun_out/${TAG}_traffic.json
echo "== C4 SQ counters"; bash tools/gpu_pmc_sq.sh ${TAG}c4 --config C4 > gpurun_out/${TAG}c4_sq.log 2>&1; python tools/pmc_sq_summary.py ${TAG}c4 > /dev/null && cp profiles/${TAG}c4_sq_counters.json gpurun_out/${TAG}_c4_sq_counters.json
rm -rf gpurun_out/sq_${TAG}c4/*/*/*.csv
echo "== power C4"; bash tools/power_probe.sh C4 > /dev/null 2>&1; mv gpurun_out/power_C4.log gpurun_out/${TAG}_power_C4.log; tail -3 gpurun_out/${TAG}_power_C4.log | cut -c1-200
echo "== power"; bash tools/power_probe.sh C2 > /dev/null 2>&1; mv gpurun_out/power_C2.log gpurun_out/${TAG}_power_C2.log; tail -3 gpurun_out/${TAG}_power_C2.log
echo "== C4 bench"; timeout -k 10 300 python bench.py --config C4 --steps 5 --warmup 2 --cpu-frames 4 2>/dev/null | grep '^{' > gpurun_out/${TAG}_C4.json.log; cut -c1-120 gpurun_out/${TAG}_C4.json.log
# (last: profiles/traffic.json has been re-stamped by the two pmc_summary runs above, so this line quotes this build's traffic)
echo "== bench (driver form)"; timeout -k 10 500 python bench.py 2>gpurun_out/${TAG}_default.err | grep '^{' > gpurun_out/${TAG}_default.json.log; cut -c1-200 gpurun_out/${TAG}_default.json.log
