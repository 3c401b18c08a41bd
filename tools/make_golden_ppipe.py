#!/usr/bin/env python3
"""Golden vectors for the sRGB post-processing pipeline from the REFERENCE itself
(dataset/fwd_ppipe.py: ppipe, psnr, find_gains; util/util.py: tensor2im), build container only:

    python3 tools/make_golden_ppipe.py

The reference module does `.cuda()` at import and inside ppipe; there is no GPU here, so
`torch.Tensor.cuda` is patched to the identity for the duration of this script.  iio / skimage / cv2 /
torchvision are absent and get the same empty stand-ins as tools/make_golden.py (none of them touches
the arithmetic captured here; SSIM -- skimage -- is therefore NOT captured).

Writes tests/golden/ppipe_*.npz and rvdd-release_amd/white_balance.json (the calibration table of
fwd_ppipe.py:12, data)."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.join(REPO, "tools"))
from make_golden import _install_standins  # noqa: E402

_install_standins()
sk = sys.modules["skimage"]
import types  # noqa: E402
skm = types.ModuleType("skimage.metrics")
skm.structural_similarity = None
sk.metrics = skm
sys.modules["skimage.metrics"] = skm
torch.Tensor.cuda = lambda self, *a, **k: self
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "dataset"))
import fwd_ppipe as R  # noqa: E402
from util.util import tensor2im  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
OUT = os.environ.get("RVDD_GOLDEN_OUT", GOLD)      # where the vectors are written (a test regenerates into a temp dir)
os.makedirs(OUT, exist_ok=True)


def run(x, seq, iso, bit_depth=8):
    """x [1,3,H,W] in [-1,1] -> what validate.py saves, then fwd_ppipe.py:128-141."""
    img = tensor2im(torch.from_numpy(x))                      # *_denoised.tif content
    n, red_gain, blue_gain = R.find_gains(seq, iso)
    rgb_gain = 1 / n
    im = img
    if bit_depth == 0:
        im = im * 4095
    elif bit_depth == 8:
        im = im / 255 * 4095
    elif bit_depth == 10:
        im = im / 1024 * 4095
    srgb = R.ppipe(im, rgb_gain, red_gain, blue_gain, iso)
    return img, srgb, srgb.round().clip(0, 255).astype(np.uint8)


seqg = np.load(os.path.join(GOLD, "seq_feat-iso3200.npz"))
rng = np.random.default_rng(5)
H, W = 40, 56
yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
ramp = np.stack([-1.2 + 2.6 * xx / (W - 1), -1.1 + 2.4 * yy / (H - 1), -1.0 + 2.0 * (xx + yy) / (H + W - 2)], 0)
ramp = (ramp + 0.01 * rng.standard_normal(ramp.shape)).astype(np.float32)          # incl. out-of-range values
cases = {
    "den3200_seq0": (seqg["denoised"][0][None], seqg["gt"][1][None], 0, 3200),
    "den3200_seq17": (seqg["denoised"][3][None], seqg["gt"][4][None], 17, 3200),
    "den12800_seq5": (seqg["denoised"][2][None], seqg["gt"][3][None], 5, 12800),
    "ramp3200_seq29": (ramp[None], np.clip(ramp, -1, 1)[None], 29, 3200),
    "ramp12800_seq11": (ramp[None], np.clip(ramp, -1, 1)[None], 11, 12800),
    "ramp_noiso_seq2": (ramp[None], np.clip(ramp, -1, 1)[None], 2, 100),           # neither ISO branch
}
for name, (x, gt, seq, iso) in cases.items():
    img, srgb, u8 = run(x.astype(np.float32), seq, iso)
    _, _, gt_u8 = run(gt.astype(np.float32), seq, iso)
    p = float(R.psnr(u8, gt_u8))
    np.savez_compressed(os.path.join(OUT, f"ppipe_{name}.npz"), x=x.astype(np.float32), gt=gt.astype(np.float32),
                        seq=seq, iso=iso, tif=img, srgb=srgb.astype(np.float32), u8=u8, gt_u8=gt_u8, psnr=p)
    print(name, "srgb range", float(srgb.min()), float(srgb.max()), "psnr", p)

# bit_depth variants of the range normalisation (fwd_ppipe.py:131-137) on one image
x = seqg["denoised"][1][None]
for bd, scale in ((0, 1.0 / 255.0), (10, 1024.0 / 255.0), (12, 4095.0 / 255.0)):
    img = tensor2im(torch.from_numpy(x)) * np.float32(scale)
    n, red_gain, blue_gain = R.find_gains(3, 3200)
    im = img * 4095 if bd == 0 else (img / 1024 * 4095 if bd == 10 else img)
    srgb = R.ppipe(im, 1 / n, red_gain, blue_gain, 3200)
    np.savez_compressed(os.path.join(OUT, f"ppipe_bitdepth{bd}.npz"), img=img.astype(np.float32), seq=3, iso=3200,
                        bit_depth=bd, srgb=srgb.astype(np.float32), u8=srgb.round().clip(0, 255).astype(np.uint8))

with open(os.path.join(REPO, "rvdd-release_amd", "white_balance.json"), "w") as f:
    json.dump({"comment": "white_balance[seq] = [[n, red_gain, blue_gain] @ISO12800, [..] @ISO3200]; rgb_gain = 1/n "
                          "(dataset/fwd_ppipe.py:12, :43-46, :117-118)", "white_balance": R.white_balance}, f)
print("wrote white_balance.json:", len(R.white_balance), "sequences")
