"""CPU oracle for the sRGB post-processing pipeline and the display-domain metrics
(reference dataset/fwd_ppipe.py).

TEST INFRASTRUCTURE ONLY: imported by tests/ and tools/ -- never by the product package.

Parity status
  * ppipe / to_uint8 / psnr_u8: PINNED -- tests/golden/ppipe_*.npz were produced by importing the
    reference's dataset/fwd_ppipe.py itself (tools/make_golden_ppipe.py: `Tensor.cuda` patched to the
    identity, iio / skimage replaced by empty stand-ins because neither is installed) and this file is
    checked against them in tests/test_ppipe.py.
  * ssim: the reference calls skimage.metrics.structural_similarity(x, y, multichannel=True,
    data_range=255) (fwd_ppipe.py:86).  scikit-image is a requirements.txt dependency with no version
    pin and is NOT installed here, so this is a restatement of its published algorithm (Wang et al.
    2004 with skimage's defaults: 7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance, border
    of (win-1)/2 cropped, mean over channels of the per-channel mean) using scipy.ndimage.uniform_filter,
    the same primitive skimage uses.  PARITY UNPINNED against skimage itself; anchored by analytic
    cases in tests/test_ppipe.py.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.ndimage import uniform_filter

# fwd_ppipe.py:15
INV_CCM = torch.tensor([[1.07955733, -0.40125771, 0.32170038],
                        [-0.15390743, 1.35677921, -0.20287178],
                        [-0.00235972, -0.55155296, 1.55391268]], dtype=torch.float32)


def tensor2im(x: torch.Tensor) -> np.ndarray:
    """util/util.py:23-49 with iT=None: [1,3,H,W] in [-1,1] -> float32 [H,W,3] in [0,255]
    (what validate.py writes to *_denoised.tif through util/visualizer.py:11-33)."""
    a = x[0].cpu().float().numpy()
    return ((np.transpose(a, (1, 2, 0)) + 1) / 2.0 * 255.0).astype(np.float32)


def normalise_bit_depth(img: np.ndarray, bit_depth: int) -> np.ndarray:
    """fwd_ppipe.py:131-137: bring the image to [0,4095]."""
    if bit_depth == 0:
        return img * 4095
    if bit_depth == 8:
        return img / 255 * 4095
    if bit_depth == 10:
        return img / 1024 * 4095
    return img


def ppipe(im: np.ndarray, rgb_gain: float, red_gain: float, blue_gain: float, iso: int) -> np.ndarray:
    """fwd_ppipe.py:48-77.  im float32 [H,W,3] in [0,4095] -> float32 sRGB x 255."""
    if iso == 3200:
        im = (im - 266) * (2305 - 245) / (3610 - 266) + 245
    if iso == 12800:
        im = (im - 268) * (2305 - 245) / (4075 - 268) + 245
    im = (im - 240) / (4095 - 240)
    t = torch.tensor(im)
    gains = torch.tensor([1.0 / (red_gain * rgb_gain), 1.0 / rgb_gain, 1.0 / (blue_gain * rgb_gain)])
    t = t / gains[None, None, :]                                   # apply_gains, :28-41
    shape = t.size()
    t = torch.tensordot(torch.reshape(t, [-1, 3]), INV_CCM, dims=[[-1], [-1]])   # apply_mat_inv_ccm, :20-26
    t = torch.reshape(t, shape)
    m = t > 10 ** -8
    t[m] = t[m] ** (1 / 2.2)
    t = 3 * t ** 2 - 2 * t ** 3
    return t.numpy() * 255


def to_uint8(srgb: np.ndarray) -> np.ndarray:
    """fwd_ppipe.py:141."""
    return srgb.round().clip(0, 255).astype(np.uint8)


def psnr_u8(img1: np.ndarray, img2: np.ndarray) -> float:
    """fwd_ppipe.py:79-84."""
    x = (np.array(img1 / 255).squeeze() - np.array(img2 / 255).squeeze()).flatten()
    return float(10 * np.log10(1 / np.mean(x ** 2)))


def ssim(x: np.ndarray, y: np.ndarray, data_range: float = 255.0, win: int = 7) -> float:
    """skimage.metrics.structural_similarity(x.astype(float), y.astype(float), multichannel=True,
    data_range=255) -- see the header."""
    x = x.astype(np.float64)
    y = y.astype(np.float64)
    NP = win * win
    cov_norm = NP / (NP - 1.0)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    pad = (win - 1) // 2
    vals = []
    for c in range(x.shape[2]):
        X, Y = x[..., c], y[..., c]
        ux, uy = uniform_filter(X, size=win), uniform_filter(Y, size=win)
        uxx, uyy, uxy = uniform_filter(X * X, size=win), uniform_filter(Y * Y, size=win), uniform_filter(X * Y, size=win)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        vals.append(S[pad:-pad, pad:-pad].mean(dtype=np.float64))
    return float(np.mean(vals))
