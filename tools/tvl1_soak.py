#!/usr/bin/env python3
"""Soak of the TV-L1 patch kernel (GPU box): the same batch of eight 640x360 pairs REPS times, then batches of random sizes and
counts twice each -- every repetition must reproduce the first one's flows and iteration counts bit for bit (a record read
half-written, a buffer reused too early or a lost update would show as a difference or as RVDD_ERR_HIP).  One JSON line."""
import json, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from rvdd_release_amd.util._ops import ops_runtime
from rvdd_release_amd import synth

reps = int(os.environ.get("REPS", "300"))
rt = ops_runtime(0)
seq = synth.make_sequence(9, 720, 1280, iso=3200, seed=21, device="cuda")
gray = seq.raw.mean(dim=1).contiguous()
a, b = gray[1:9].contiguous(), gray[0:8].contiguous()
ref, it_ref = rt.tvl1flow_batch(a, b, want_iterations=True)
bad = 0
t0 = time.perf_counter()
for r in range(reps):
    f, it = rt.tvl1flow_batch(a, b, want_iterations=True)
    if list(it) != list(it_ref) or not torch.equal(f, ref):
        bad += 1
torch.cuda.synchronize()
el = time.perf_counter() - t0
rng = np.random.default_rng(5)
sizes_bad = 0
cases = 0
for _ in range(int(os.environ.get("SIZES", "40"))):
    h, w, n = int(rng.integers(16, 300)), int(rng.integers(16, 500)), int(rng.integers(1, 10))
    x = torch.from_numpy(rng.random((n + 1, h, w), dtype=np.float32)).cuda()
    x = torch.nn.functional.avg_pool2d(x[None], 5, 1, 2)[0].contiguous()      # smooth enough to have a flow
    try:
        f1, i1 = rt.tvl1flow_batch(x[1:].contiguous(), x[:-1].contiguous(), want_iterations=True)
        f2, i2 = rt.tvl1flow_batch(x[1:].contiguous(), x[:-1].contiguous(), want_iterations=True)
    except RuntimeError as e:
        if "skinny" in str(e):
            continue
        raise
    cases += 1
    if list(i1) != list(i2) or not torch.equal(f1, f2) or not torch.isfinite(f1).all():
        sizes_bad += 1
print(json.dumps({"what": "TV-L1 patch kernel soak", "repetitions_of_8_pairs_640x360": reps, "repetitions_that_differed": bad,
                  "ms_per_flow": round(1e3 * el / reps / 8, 3), "iterations": [int(i) for i in it_ref],
                  "random_size_batches": cases, "random_size_batches_that_differed": sizes_bad}))
rt.close()
