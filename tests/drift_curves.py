#!/usr/bin/env python3
"""Per-frame max-abs difference between the HIP path and the CPU oracle over WHOLE sequences (30 frames for the nets
of C2 / C3 / C4, 90 for C5), B = 2 at 180x320 -- the curves behind tests/test_gpu_sequences.py::test_whole_sequence_vs_oracle.
usage (GPU box, repo root): python tests/drift_curves.py > gpurun_out/drift_curves.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rvdd_oracle as O  # noqa: E402  (kept under tests/: the oracle is the checker here, nothing under oracle/ is used outside tests/, smoke() and the bench baseline)
from safetensors.torch import load_file  # noqa: E402
from rvdd_release_amd import synth  # noqa: E402
from rvdd_release_amd.runtime import RvddRuntime  # noqa: E402

CASES = [("C2", "convunet+feat", "recurrent-convunet+feat-iso3200", 0, 3200, 30),
         ("C3", "convunet+feat", "recurrent-convunet+feat-future-iso12800", 1, 12800, 30),
         ("C4", "next+feat", "recurrent-ConvNeXtUnet+feat-future-iso3200", 1, 3200, 30),
         ("C5", "convunet+feat", "recurrent-convunet+feat-iso3200", 0, 3200, 90)]
out = {"what": "max |hip - oracle| per output frame, sequence 0 of a B = 2 batch at 180x320; task PSNR of both on the last frame",
       "cases": {}}
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
for cfg, arch, stem, fut, iso, T in CASES:
    sd = load_file(os.path.join(ROOT, "weights", stem + ".safetensors"))
    B, H, W = 2, 180, 320
    seqs = [synth.make_sequence(T, H, W, iso=iso, seed=8000 + 10 * int(cfg[1]) + b) for b in range(B)]
    rt = RvddRuntime(arch, fut, B, H, W, 0)
    rt.load_state_dict(sd)
    raw = torch.stack([s.raw for s in seqs], 1).cuda()
    fp = torch.stack([s.flow_prev for s in seqs], 1).cuda()
    fn = torch.stack([s.flow_next for s in seqs], 1).cuda()
    n_out = T - 1 - fut
    got = torch.empty(n_out, B, 3, H, W, device="cuda")
    for t in range(1, T - fut):
        rt.step(raw[t - 1] if t == 1 else None, raw[t], raw[t + 1] if fut else None, fp[t], fn[t] if fut else None, out=got[t - 1])
    got = got.cpu()
    want = O.RecurrentOracle(sd, future=fut).run_sequence(seqs[0].raw, seqs[0].flow_prev, seqs[0].flow_next)
    curve = [float((got[k, 0] - want[k]).abs().max()) for k in range(n_out)]
    gt = seqs[0].gt[n_out][None]
    out["cases"][cfg] = {"frames": n_out, "max_abs_per_frame": [float(f"{v:.3e}") for v in curve], "worst": max(curve),
                         "task_psnr_last_hip": O.psnr(got[-1, 0][None], gt), "task_psnr_last_oracle": O.psnr(want[-1][None], gt)}
    rt.close()
print(json.dumps(out, indent=1))
