#!/bin/bash
# GPU box: the fused-upsample conv kernel with its code moved by 4 x PAD bytes (conv3x3h.hip, -DRVDD_UPS_PAD), each build checked against the two-kernel
# form over sequences at a bench size (tools/fused_upsample_check.py) and run to run (tools/determinism_soak.py).  Round 6's defect showed at some
# alignments of the code and not at others (profiles/r06s_upsample_nondeterminism.md): a clean result at ONE alignment says little.
#   build here first:  bash tools/ups_layout_matrix.sh build [PAD ...]      (librvdd_hip_pad<PAD>.so beside the library; they travel with gpurun)
#   on the GPU box:    bash tools/ups_layout_matrix.sh run [PAD ...]
mode=$1; shift
pads=${@:-0 1 2 3 5 8 16}
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
if [ "$mode" = build ]; then
  mkdir -p tools/scratch
  for k in $pads; do
    (cd rvdd-release_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DRVDD_UPS_PAD=$k -c conv3x3h.hip -o ../../tools/scratch/conv3x3h_pad$k.o &&
      hipcc --offload-arch=gfx950 -shared -fPIC conv3x3.o ../../tools/scratch/conv3x3h_pad$k.o wino3x3.o convnext.o prestage.o tvl1.o srgb.o runtime.o -o ../librvdd_hip_pad$k.so) && echo "built pad $k"
  done
else
  cp rvdd-release_amd/librvdd_hip.so /tmp/keep_layout.so
  for k in $pads; do
    cp rvdd-release_amd/librvdd_hip_pad$k.so rvdd-release_amd/librvdd_hip.so
    echo "== pad $k"
    timeout -k 10 300 python tools/fused_upsample_check.py C2 20 3 2>&1 | grep "^{" | cut -c1-400
    timeout -k 10 300 python tools/determinism_soak.py C2 ${SOAK_REPS:-6} 60 2>&1 | grep "^{" | cut -c1-400
  done
  cp /tmp/keep_layout.so rvdd-release_amd/librvdd_hip.so
fi
