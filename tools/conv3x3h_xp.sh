#!/bin/bash
# The "what bounds a tile" table of conv3x3h_kernel: tools/conv3x3h_xp_patch.py's instrumented copy, one build per switch set.
#   bash tools/conv3x3h_xp.sh "0 1 2 4 8 3 6 14 15 31 47 63" [full]     (full: the per-wave stamp table too)
set -o pipefail
mkdir -p tools/scratch
python3 tools/conv3x3h_xp_patch.py tools/scratch/conv3x3h_xp.hip >/dev/null || exit 1
for xp in $1; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -DRVDD_STAMPS -DRVDD_CONV_GROUPS2 -DRVDD_XP=$xp '-DCONV_SRC="scratch/conv3x3h_xp.hip"' -Wno-unused-value \
        -Irvdd-release_amd/csrc -Itools tools/conv3x3h_bench.hip -o /tmp/c3hb_$xp 2>/dev/null || { echo "build $xp failed"; continue; }
  if [ "$2" = full ]; then /tmp/c3hb_$xp 4 720 1280 1; else /tmp/c3hb_$xp 4 720 1280 1 | head -1; fi
done
