#!/usr/bin/env python3
"""Run-to-run determinism soak: the same B sequences of T frames through one runtime REPS times, every repetition compared bit for bit
with the first (frames and recurrent features).  Reports every repetition that differs: first frame / slot, element count, rows, columns.
usage (GPU box): python tools/determinism_soak.py [config C1|C2|C3|C4] [REPS] [T] [B]      -> one JSON line
(B = 1 with C1 is the shape that takes the one-kernel pre-stage and the output-channel split of the convs)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from safetensors.torch import load_file
from rvdd_release_amd import synth
from rvdd_release_amd.runtime import RvddRuntime

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
arch, stem, fut, iso, H, W, T, B, _ = bench.CONFIGS[cfg]
T = int(sys.argv[3]) if len(sys.argv) > 3 else 12
B = int(sys.argv[4]) if len(sys.argv) > 4 else B
sd = load_file(os.path.join(bench.REPO, "weights", stem + ".safetensors"))
seqs = [synth.make_sequence(T, H, W, iso=iso, seed=4000 + b, device="cuda") for b in range(B)]
raw = torch.stack([s.raw for s in seqs], 1).contiguous()
fp = torch.stack([s.flow_prev for s in seqs], 1).contiguous()
fn = torch.stack([s.flow_next for s in seqs], 1).contiguous() if fut else None
del seqs
rt = RvddRuntime(arch, fut, B, H, W, 0)
rt.load_state_dict(sd)
n_out = T - 1 - fut


def run():
    outs = torch.empty(n_out, B, 3, H, W, device="cuda")
    bench.advance(rt, raw, fp, fn, outs, T, fut, False)
    return outs, rt.get_state()[1]


ref, ref_feat = run()
bad = []
for r in range(1, reps):
    o, f = run()
    if not torch.equal(o, ref) or (f is not None and not torch.equal(f, ref_feat)):
        d = (o - ref).abs()
        pairs = [(t, b) for t in range(n_out) for b in range(B) if bool((d[t, b] > 0).any())]
        t0, b0 = pairs[0] if pairs else (-1, -1)
        info = {"rep": r, "pairs": len(pairs), "first": [t0, b0]}
        if pairs:
            ys = torch.nonzero(d[t0, b0] > 0)
            info.update(elements=int((d[t0, b0] > 0).sum()), max_abs=float(d[t0, b0].max()), rows=[int(ys[:, 1].min()), int(ys[:, 1].max())],
                        cols=[int(ys[:, 2].min()), int(ys[:, 2].max())])
        bad.append(info)
    del o, f
print(json.dumps({"config": cfg, "reps": reps, "frames": T, "batch": B, "differing_repetitions": bad, "finite": bool(torch.isfinite(ref).all())}))
rt.close()
