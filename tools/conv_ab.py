#!/usr/bin/env python3
"""A/B of conv3x3 code variants in ONE process, interleaved rounds (guide rule 24)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safetensors.torch import load_file
from rvdd_release_amd.runtime import RvddRuntime

H, W, B = 720, 1280, int(os.environ.get("B", "1"))
variants = [int(v) for v in os.environ.get("VARIANTS", "0,3").split(",")]
rt = RvddRuntime("convunet+feat", 0, B, H, W, 0)
rt.load_state_dict(load_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "weights", "recurrent-convunet+feat-iso3200.safetensors")))
# fill the level-0 input map with random data (zeros would inflate the clock)
x = torch.randn(B, 6, H, W, device="cuda"); f = torch.randn(B, 48, H, W, device="cuda").relu()
rt.unet_forward(x, f)
flops = lambda lvl: 2 * 9 * 48 * 48 * B * (H >> lvl) * (W >> lvl)
for lvl in (0, 1, 2, 3):
    res = {v: [] for v in variants}
    for r in range(7):
        for v in variants:
            res[v].append(rt.debug_conv_bench(v, lvl, 20 if lvl < 2 else 100))
    for v in variants:
        med, mn = statistics.median(res[v]), min(res[v])
        print(f"level {lvl} variant {v}: median {med*1e3:8.1f} us  min {mn*1e3:8.1f} us  -> {flops(lvl)/med/1e9:6.1f} TFLOP/s (median)")
