#!/usr/bin/env python3
"""The decoder's fused upsample (conv3x3h_kernel with the bilinear x2 in its halo fetch) against the two-kernel form (upsample2x_kernel, then the
same conv) over whole sequences at a bench size: B sequences of T frames through two runtimes that differ in the option fuse_upsample only, every
output frame and the recurrent features compared bit for bit, REPS times (fresh runtimes each time: the launches in front of each kernel differ
from a warmed-up loop's).  One JSON line.  The small shapes of tests/test_gpu_parity.py::test_fused_upsample_equals_upsample_then_conv give every
workgroup one tile; the defect of round 6 (profiles/r06s_upsample_nondeterminism.md) needed a workgroup's SECOND tile to be a border tile.
usage (GPU box): python tools/fused_upsample_check.py [config C2|C3|C4] [T] [REPS]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from safetensors.torch import load_file
from rvdd_release_amd import synth
from rvdd_release_amd.runtime import RvddRuntime


def compare(cfg="C2", T=12, reps=1):
    arch, stem, fut, iso, H, W, _, B, _ = bench.CONFIGS[cfg]
    sd = load_file(os.path.join(bench.REPO, "weights", stem + ".safetensors"))
    seqs = [synth.make_sequence(T, H, W, iso=iso, seed=4100 + b, device="cuda") for b in range(B)]
    raw = torch.stack([s.raw for s in seqs], 1).contiguous()
    fp = torch.stack([s.flow_prev for s in seqs], 1).contiguous()
    fn = torch.stack([s.flow_next for s in seqs], 1).contiguous() if fut else None
    del seqs
    n_out = T - 1 - fut
    bad = []
    for rep in range(reps):
        res = []
        for fused in (1, 0):
            rt = RvddRuntime(arch, fut, B, H, W, 0)
            rt.set_option("fuse_upsample", fused)
            rt.load_state_dict(sd)
            outs = torch.empty(n_out, B, 3, H, W, device="cuda")
            bench.advance(rt, raw, fp, fn, outs, T, fut, False)
            res.append((outs, rt.get_state()[1]))
            rt.close()
        (a, fa), (b, fb) = res
        if not torch.equal(a, b) or (fa is not None and not torch.equal(fa, fb)):
            d = (a - b).abs()
            pairs = [(t, s) for t in range(n_out) for s in range(B) if bool((d[t, s] > 0).any())]
            t0, s0 = pairs[0]
            ys = torch.nonzero(d[t0, s0] > 0)
            bad.append({"rep": rep, "pairs": len(pairs), "first": [t0, s0], "elements": int((d[t0, s0] > 0).sum()), "max_abs": float(d[t0, s0].max()),
                        "rows": [int(ys[:, 1].min()), int(ys[:, 1].max())], "cols": [int(ys[:, 2].min()), int(ys[:, 2].max())]})
        del res, a, b, fa, fb
    return {"config": cfg, "frames": T, "batch": B, "reps": reps, "differing": bad}


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    print(json.dumps(compare(cfg, T, reps)))
