#!/usr/bin/env python3
"""GPU box: one-off sweep of small odd frame shapes through the C ABI against the CPU oracle (two frame-steps each),
both conv kernels for convunet, and ConvNeXtUnet.  Prints one line per case; exit code 1 if any exceeds 1e-4."""
import os, sys, itertools
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from rvdd_release_amd import synth
from rvdd_release_amd.runtime import RvddRuntime
from safetensors.torch import load_file
sys.path.insert(0, os.path.join(REPO, "oracle"))
import rvdd_oracle as O

SHAPES = [(16, 16), (18, 34), (22, 130), (130, 22), (50, 66), (98, 30), (16, 258), (64, 64), (24, 40), (34, 18), (258, 16)]
CASES = [("convunet+feat", "recurrent-convunet+feat-iso3200", 0, (0, 1)),
         ("convunet", "recurrent-convunet-future-iso3200", 1, (0, 1)),
         ("next+feat", "recurrent-ConvNeXtUnet+feat-future-iso3200", 1, (None,)),
         ("next", "recurrent-ConvNeXtUnet-iso3200", 0, (None,))]
bad = 0
for (arch, stem, fut, kernels), (H, W), B in itertools.product(CASES, SHAPES, (1, 3)):
    sd = load_file(os.path.join(REPO, "weights", stem + ".safetensors"))
    T = 3 + fut
    seqs = [synth.make_sequence(T, H, W, iso=3200, seed=900 + b) for b in range(B)]
    want = [O.RecurrentOracle(sd, future=fut).run_sequence(s.raw, s.flow_prev, s.flow_next) for s in seqs]
    raw = torch.stack([s.raw for s in seqs], 0).cuda()
    fp = torch.stack([s.flow_prev for s in seqs], 0).cuda()
    fn = torch.stack([s.flow_next for s in seqs], 0).cuda()
    for k in kernels:
        rt = RvddRuntime(arch, fut, B, H, W, 0)
        if k is not None:
            rt.set_option("conv_kernel", k)
        rt.load_state_dict(sd)
        worst = 0.0
        for t in range(1, T - fut):
            out = rt.step(raw[:, t - 1], raw[:, t], raw[:, t + 1] if fut else None, fp[:, t], fn[:, t] if fut else None).cpu()
            for b in range(B):
                worst = max(worst, float((out[b] - want[b][t - 1]).abs().max()))
        rt.close()
        flag = "" if worst < 1e-4 else "  <-- FAIL"
        bad += worst >= 1e-4
        print(f"{arch:14s} fut={fut} kernel={k} B={B} {H}x{W}: max-abs {worst:.2e}{flag}", flush=True)
sys.exit(1 if bad else 0)
