#!/bin/bash
# GPU box: tools/determinism_soak.py over builds of the library and environment switches: bash tools/determinism_bisect.sh REPS variant[:ENV=VAL,...] ...
REPS=$1; shift
cd ${GRAFT_REPO_ROOT:-$PWD}
cp rvdd-release_amd/librvdd_hip.so /tmp/keep.so
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  cp rvdd-release_amd/librvdd_hip_$v.so rvdd-release_amd/librvdd_hip.so
  echo "== $spec"; env $envs timeout -k 10 400 python tools/determinism_soak.py C2 $REPS 90 2>&1 | grep "^{" | cut -c1-600
done
cp /tmp/keep.so rvdd-release_amd/librvdd_hip.so
