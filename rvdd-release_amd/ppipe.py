"""dataset/fwd_ppipe.py of the reference on the HIP runtime: camera-linear RGB -> display sRGB
(`ppipe`), the white-balance table (`find_gains`) and the display-domain metrics (`psnr`, `ssim`).

The arithmetic runs in librvdd_hip.so (`rvdd_ppipe`, `rvdd_srgb_metrics`); inputs that arrive as
NumPy arrays are uploaded, results come back in the type they came in (tensor in -> tensor out)."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from .util._ops import ops_runtime

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "white_balance.json")) as _f:
    # white_balance[seq] = [[n, red_gain, blue_gain] at ISO 12800, [...] at ISO 3200]  (fwd_ppipe.py:12)
    WHITE_BALANCE = json.load(_f)["white_balance"]


def find_gains(seq: int, iso: int):
    """fwd_ppipe.py:43-46.  Returns [n, red_gain, blue_gain]; the caller passes rgb_gain = 1/n."""
    if iso == 3200:
        return WHITE_BALANCE[seq][1]
    return WHITE_BALANCE[seq][0]


def _to_dev(x, device, dtype):
    t = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))
    return t.to(torch.device("cuda", device), dtype)


def ppipe(im, rgb_gain, red_gain, blue_gain, iso, bit_depth: int = 12, layout: str = None,
          want_float: bool = False, device: int = 0):
    """fwd_ppipe.py:48-77 with the range normalisation (:131-137) and uint8 conversion (:141) fused.
    `im`: one image [H,W,3] / [3,H,W] or a batch; `bit_depth` as the reference's --bit_depth (12 = the
    image is already in [0,4095], which is what the reference's ppipe() itself takes), -1 = network
    output in [-1,1].  Returns uint8 [..,H,W,3] (and the float image x255 with `want_float`)."""
    was_np = not torch.is_tensor(im)
    t = _to_dev(im, device, torch.float32)
    single = t.dim() == 3
    if single:
        t = t[None]
    if layout is None:                                     # channels-last (the reference's images) unless only dim -3 is 3
        layout = "nchw" if (t.shape[1] == 3 and t.shape[3] != 3) else "hwc"
    out = ops_runtime(t.device.index or 0).ppipe(t, rgb_gain, red_gain, blue_gain, iso, bit_depth, layout, want_float)
    outs = out if want_float else (out,)
    if single:
        outs = tuple(o[0] for o in outs)
    if was_np:
        outs = tuple(o.cpu().numpy() for o in outs)
    return outs if want_float else outs[0]


def srgb_metrics(a, b, device: int = 0):
    """(psnr[n], ssim[n]) of uint8 [n,H,W,3] image batches."""
    ta, tb = _to_dev(a, device, torch.uint8), _to_dev(b, device, torch.uint8)
    return ops_runtime(ta.device.index or 0).srgb_metrics(ta, tb)


def psnr(img1, img2, device: int = 0) -> float:
    """fwd_ppipe.py:79-84 on two uint8 [H,W,3] images."""
    return srgb_metrics(_to_dev(img1, device, torch.uint8)[None], _to_dev(img2, device, torch.uint8)[None])[0][0]


def ssim(x, y, device: int = 0) -> float:
    """fwd_ppipe.py:86 on two uint8 [H,W,3] images."""
    return srgb_metrics(_to_dev(x, device, torch.uint8)[None], _to_dev(y, device, torch.uint8)[None])[1][0]


def main(argv=None):
    """The script part of dataset/fwd_ppipe.py (:92-163): every <result_folder>/<seq>/<frame>_denoised.tif
    -> <frame>_processed_pipeline.png, PSNR / SSIM against <validation_path>/gt_RGB_iso<ISO>/<seq>/<frame>.png,
    PSNR.txt / SSIM.txt with the averages.  Returns (average_psnr, average_ssim)."""
    import argparse
    from .library import iio_read, iio_write
    p = argparse.ArgumentParser(description="Compute the forward pipeline")
    p.add_argument("--validation_path", type=str, default="Path_to_validation_dataset")
    p.add_argument("--result_folder", type=str, default="%03d")
    p.add_argument("--videos", type=str, default='')
    p.add_argument("--first", type=int, default=3)
    p.add_argument("--last", type=int, default=264)
    p.add_argument("--step", type=int, default=3)
    p.add_argument("--bit_depth", type=int, default=8)
    p.add_argument("--ISO", type=int, default=3200)
    opt = p.parse_args(argv)
    seqs = list(range(30)) if opt.videos == '' else [int(s) for s in opt.videos.split(',')]
    list_psnr, list_ssim = [], []
    with open(os.path.join(opt.result_folder, "PSNR.txt"), 'w') as plot_psnr, \
            open(os.path.join(opt.result_folder, "SSIM.txt"), 'w') as plot_ssim:
        for seq in seqs:
            n, red_gain, blue_gain = find_gains(seq, opt.ISO)
            for i in range(opt.first, opt.last + opt.step, opt.step):
                img = iio_read(os.path.join(opt.result_folder, "{:03d}/{:08d}_denoised.tif".format(seq, i)))
                assert img.shape[-1] == 3, "The data should have 3 channels."
                sRGB = ppipe(img.astype(np.float32), 1 / n, red_gain, blue_gain, opt.ISO, bit_depth=opt.bit_depth, layout="hwc")
                iio_write(sRGB, os.path.join(opt.result_folder, "{:03d}/{:08d}_processed_pipeline.png".format(seq, i)))
                gt = iio_read(os.path.join(opt.validation_path, "gt_RGB_iso{:1d}/{:03d}/{:08d}.png".format(opt.ISO, seq, i)))
                ps, ss = srgb_metrics(sRGB[None], gt[None])
                list_psnr.append(ps[0])
                list_ssim.append(ss[0])
                plot_psnr.write(str(ps[0]) + '\n')
                plot_ssim.write(str(ss[0]) + '\n')
        average_psnr, average_ssim = float(np.mean(list_psnr)), float(np.mean(list_ssim))
        plot_psnr.write("\n\n###  Average: {:4.2f} dB  ###".format(average_psnr))
        plot_ssim.write("\n\n###  Average: {:4.3f}  ###".format(average_ssim))
    print("Average PSNR: {:4.2f}".format(average_psnr))
    print("Average SSIM: {:4.3f}".format(average_ssim))
    return average_psnr, average_ssim


if __name__ == "__main__":
    main()
