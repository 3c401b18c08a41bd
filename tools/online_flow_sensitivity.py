#!/usr/bin/env python3
"""validate.py's online-flow loop (flows recomputed by TV-L1 from the previous DENOISED frame) amplifies rounding
differences of the denoiser: max |frame - oracle's frame| per output frame for the conv kernel choices, and the flow
difference behind it.  GPU box:  python tools/online_flow_sensitivity.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import torch
import test_validate as TV

def run(conv):
    if conv: os.environ["RVDD_CONV"] = conv
    else: os.environ.pop("RVDD_CONV", None)
    from rvdd_release_amd import synth
    from rvdd_release_amd.models import create_model
    from rvdd_release_amd.options import make_opt
    from rvdd_release_amd.validate import compute_validation
    sd = TV.load_weights(TV.STEM)
    seqs = [synth.make_sequence(4, 96, 128, iso=3200, seed=60 + v) for v in range(2)]
    opt = make_opt(netDenoiser=TV.NET, feature_rec=True, future_patch_depth=0, path2epoch=os.path.join(TV.WEIGHTS, TV.STEM),
                   gpu_ids=[0], val_flow_from_denoised=True)
    model = create_model(opt); model.setup(opt); opt.isTrain = False; model.isTrain = False
    got = []
    compute_validation(model, TV._dataset(seqs), opt, on_frame=lambda i, d, vis, l: got.append(vis["denoised"][0].cpu()))
    return got, seqs, sd, opt

if __name__ == "__main__":
    res = {}
    for conv in ("", "f32", "winograd"):
        res[conv or "default"], seqs, sd, opt = run(conv)
    want, _, _ = TV._oracle_loop(sd, seqs, True, opt.lambda_L1)
    for k, got in res.items():
        d = [(g - w).abs() for g, w in zip(got, want)]
        print(f"{k:9s} vs oracle: max per frame", [f"{float(x.max()):.2e}" for x in d], " pixels > 2e-3:", [int((x > 2e-3).sum()) for x in d])
    for a, b in (("default", "f32"), ("f32", "winograd")):
        print(f"{a} vs {b}: max per frame", [f"{float((x - y).abs().max()):.2e}" for x, y in zip(res[a], res[b])])
    # the flow that frame 2 of video 0 is warped with, from the two first-frame outputs (1e-6 apart)
    from rvdd_release_amd.util._ops import ops_runtime
    from rvdd_release_amd.util.Hamilton_Adam_demo import HamiltonAdam
    rt = ops_runtime(0)
    s = seqs[0]
    tgt = ((s.raw[2].cuda() + 1) / 2).mean(0).contiguous()
    flows = {}
    for k in ("default", "f32", "winograd"):
        mv = ((HamiltonAdam('gbrg').remosaick(res[k][0][None].cuda())[0] + 1) / 2).mean(0).contiguous()
        flows[k] = rt.tvl1flow(tgt, mv, want_iterations=True)
        print(f"{k:9s}: TV-L1 iterations {flows[k][1]}")
    for a, b in (("default", "f32"), ("f32", "winograd")):
        d = (flows[a][0] - flows[b][0]).abs()
        print(f"flow {a} vs {b}: max {float(d.max()):.3e} px, mean {float(d.mean()):.3e}, > 1e-3 px: {int((d > 1e-3).sum())}")
