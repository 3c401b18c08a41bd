"""util/flow_utils.py of the reference, hot-path functions only, on the HIP runtime."""
from __future__ import annotations

import torch

from ._ops import dev_index, ops_runtime


def warp(x, flow, interp="bicubic"):
    """``warp(x, flow, interp) -> (y, mask)`` (util/flow_utils.py:70-102).

    Only ``interp="bicubic"`` is built (every hot-path caller passes it,
    models/recurrent_model.py:151,154,297).  The mask is computed as the
    reference does (:95-96) and, like there, returned as a CPU FloatTensor."""
    if interp != "bicubic":
        raise NotImplementedError(f"rvdd warp: interp={interp!r} is not built (bicubic only)")
    B, C, H, W = x.shape
    flow = flow.to(x.device)
    y = ops_runtime(dev_index(x)).warp(x.float(), flow.float())
    yy, xx = torch.meshgrid(torch.arange(H, device=x.device), torch.arange(W, device=x.device), indexing="ij")
    gx = 2.0 * (xx[None].float() + flow[:, 0]) / (W - 1) - 1.0
    gy = 2.0 * (yy[None].float() + flow[:, 1]) / (H - 1) - 1.0
    mask = (gx >= -1) * (gx <= 1) * (gy >= -1) * (gy <= 1)
    return y, mask.unsqueeze(1).type(torch.FloatTensor)


def upsample_factor_2(downsampled_batch, multiply_by=1.):
    """util/flow_utils.py:159-174: [...,C,H,W] -> [...,C,2H,2W], bilinear align_corners=True."""
    return ops_runtime(dev_index(downsampled_batch)).upsample_factor_2(downsampled_batch.float(), multiply_by)


def single_warp(iio_img_like, np_flow, interpolation="bicubic", givemask=False):
    """util/flow_utils.py:104-121: warp one [H,W,C] image by a [H,W,2] flow, NumPy in / NumPy out."""
    import numpy as np
    img = torch.from_numpy(np.ascontiguousarray(np.asarray(iio_img_like, np.float32).transpose(2, 0, 1)))[None].cuda()
    flow = torch.from_numpy(np.ascontiguousarray(np.asarray(np_flow, np.float32).transpose(2, 0, 1)))[None].cuda()
    warped, mask = warp(img, flow, interpolation)
    out = warped.cpu().numpy().squeeze().transpose(1, 2, 0)
    return (out, mask) if givemask else out


def compute_flow(iio_img1, iio_img2, flow_type='tvl1'):
    """util/flow_utils.py:124-133."""
    from ..library import CPPbridge
    if flow_type != 'tvl1':
        raise TypeError(f"Unknown flow type {flow_type}")
    return CPPbridge('./build/libBridge.so').TVL1_flow(iio_img2, iio_img1)


def compute_flow_and_warp(iio_img1, iio_img2, flow_type='tvl1', interpolation='bicubic', iio_flow_img1=None):
    """util/flow_utils.py:136-156: flow from img2 to (flow_)img1 with TV-L1, then img1 warped onto img2."""
    from ..library import CPPbridge
    if iio_flow_img1 is None:
        iio_flow_img1 = iio_img1
    if flow_type != 'tvl1':
        raise TypeError(f"Unknown flow type {flow_type}")
    flow = CPPbridge('./build/libBridge.so').TVL1_flow(iio_img2, iio_flow_img1)
    warped, undef_mask = single_warp(iio_img1, flow, interpolation, givemask=True)
    return warped, undef_mask, flow
