#!/usr/bin/env python3
"""sRGB post-processing + display metrics throughput on 1280x720 frames:
HIP path (rvdd_ppipe, rvdd_srgb_metrics) against its HBM roofline, with the CPU oracle timed beside it.
One JSON line.  ppipe: 12 B read + 3 B written per pixel; metrics: 2 x 3 B read per pixel."""
import json, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
from rvdd_release_amd.util._ops import ops_runtime
from rvdd_release_amd.ppipe import find_gains

B, H, W, iters = int(os.environ.get("BATCH", "4")), 720, 1280, 50
gen = torch.Generator().manual_seed(1)
x = (torch.rand(B, 3, H, W, generator=gen) * 2 - 1).cuda()
n, red, blue = find_gains(7, 3200)
rt = ops_runtime(0)
u8 = rt.ppipe(x, 1 / n, red, blue, 3200, -1)
gt = rt.ppipe((x + 0.02 * torch.randn_like(x)).clamp(-1, 1), 1 / n, red, blue, 3200, -1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    rt.ppipe(x, 1 / n, red, blue, 3200, -1)
e1.record(); torch.cuda.synchronize()
ms_p = e0.elapsed_time(e1) / iters
t0 = time.perf_counter()
for _ in range(iters):
    ps, ss = rt.srgb_metrics(u8, gt)                 # synchronous (returns host doubles)
ms_m = 1e3 * (time.perf_counter() - t0) / iters
px = B * H * W
out = {"metric": "sRGB ppipe frames/sec, 1280x720", "batch": B,
       "ppipe_frames_per_s": round(B / ms_p * 1e3, 1), "ppipe_ms_per_launch": round(ms_p, 4),
       "ppipe_roofline": {"bound": "hbm", "achieved": round(px * 15 / ms_p / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                          "frac": round(px * 15 / ms_p / 1e6 / 8000.0, 4)},
       "metrics_frames_per_s": round(B / ms_m * 1e3, 1), "metrics_ms_per_call": round(ms_m, 4),
       "psnr": ps[0], "ssim": ss[0]}
import ppipe_oracle as P
a = x[:1].cpu()
t0 = time.perf_counter()
img = P.tensor2im(a)
srgb = P.ppipe(P.normalise_bit_depth(img, 8), 1 / n, red, blue, 3200)
cu8 = P.to_uint8(srgb)
t1 = time.perf_counter()
cp, cs = P.psnr_u8(cu8, gt[0].cpu().numpy()), P.ssim(cu8, gt[0].cpu().numpy())
t2 = time.perf_counter()
d = np.abs(cu8.astype(int) - u8[0].cpu().numpy().astype(int))
out.update({"cpu_oracle_ppipe_frames_per_s": round(1 / (t1 - t0), 2), "cpu_oracle_metrics_frames_per_s": round(1 / (t2 - t1), 2),
            "u8_mismatch_fraction": float((d != 0).mean()), "u8_max_diff": int(d.max()),
            "psnr_diff": abs(cp - ps[0]), "ssim_diff": abs(cs - ss[0])})
print(json.dumps(out))
