"""library.CPPbridge of the reference (library.py:143-175) on the HIP runtime: the TV-L1 optical flow
that the reference computes with its native libBridge.so (libBridge.cpp:44-163)."""
from __future__ import annotations

import numpy as np
import torch

from .util._ops import ops_runtime


# skimage.color.rgb2gray: Y = 0.2125 R + 0.7154 G + 0.0721 B (ITU-R 709 luma on linear values), integer images
# scaled to [0,1] first (img_as_float).  scikit-image is not installed here and the reference does not pin it:
# restated from its published definition, parity with skimage itself unpinned.
_RGB2GRAY = (0.2125, 0.7154, 0.0721)


def _gray(a: torch.Tensor, dev) -> torch.Tensor:
    """[h,w,c] image -> the [h,w] float32 plane libBridge's tvl1flow receives (library.py:160-168)."""
    if a.dim() != 3 or a.shape[2] not in (1, 3, 4):
        raise AssertionError("TVL1_flow: images must be [h,w,c] with c = 1 (gray), 3 (RGB) or 4 (packed raw)")
    if a.shape[2] == 3:
        x = a.to(dev, torch.float64)
        if not a.dtype.is_floating_point:
            x = x / float(torch.iinfo(a.dtype).max)
        w = torch.tensor(_RGB2GRAY, dtype=torch.float64, device=dev)
        return (x @ w).to(torch.float32).contiguous()
    return a.to(dev, torch.float32).mean(dim=2).contiguous()


class CPPbridge(object):
    """Same constructor and method as the reference's ctypes bridge; `libpath` is accepted and
    ignored (the flow runs in librvdd_hip.so on `device`)."""

    def __init__(self, libpath=None, device: int = 0):
        self.device = device
        self.rt = ops_runtime(device)

    def TVL1_flow(self, Im1, Im2):
        """Im1, Im2: [h,w,c] images (numpy or torch; c = 1, 3 or 4).  4-channel raw frames are reduced by the
        channel mean (library.py:165-167), RGB images by the luminance of skimage.color.rgb2gray
        (library.py:160-162; see `_gray`).
        Returns the flow as a float32 numpy array [h,w,2] such that Im2(x + flow) ~ Im1(x)."""
        a = Im1 if torch.is_tensor(Im1) else torch.as_tensor(np.asarray(Im1))
        b = Im2 if torch.is_tensor(Im2) else torch.as_tensor(np.asarray(Im2))
        if a.shape != b.shape:
            raise AssertionError("Both images Im1 and Im2 are supposed to share same size")
        dev = torch.device("cuda", self.device)
        flow = self.rt.tvl1flow(_gray(a, dev), _gray(b, dev))
        return flow.permute(1, 2, 0).contiguous().cpu().numpy()

    def TVL1_flow_batch(self, Im1s, Im2s):
        """Several pairs of the same size at once (not in the reference: its dataset code calls TVL1_flow in a
        loop, data/base_dataset.py:134-249): lists of [h,w,c] images -> list of [h,w,2] flows, each identical to
        what TVL1_flow returns for that pair."""
        if len(Im1s) != len(Im2s):
            raise AssertionError("TVL1_flow_batch: the two lists differ in length")
        if len(Im1s) == 0:
            return []
        dev = torch.device("cuda", self.device)

        def gray(ims):
            return torch.stack([_gray(im if torch.is_tensor(im) else torch.as_tensor(np.asarray(im)), dev)
                                for im in ims], 0).contiguous()
        flows = self.rt.tvl1flow_batch(gray(Im1s), gray(Im2s))
        return [f.permute(1, 2, 0).contiguous().cpu().numpy() for f in flows]


# ---- file helpers of library.py (host side; no arithmetic beyond the [0,1] scaling) -----------------
import fnmatch  # noqa: E402
import os  # noqa: E402

from . import tiffio  # noqa: E402


def iio_read(path):
    """library.py:75-77.  TIFF through rvdd's own reader, 8-bit formats (png, jpg) through Pillow;
    always [H,W,C] like iio."""
    if path.lower().endswith(('.tif', '.tiff')):
        return tiffio.read(path)
    from PIL import Image
    return np.atleast_3d(np.array(Image.open(path)))


def iio_write(arr, path):
    """library.py:71-73 (note the argument order)."""
    if path.lower().endswith(('.tif', '.tiff')):
        return tiffio.write(path, arr)
    from PIL import Image
    a = np.asarray(arr)
    Image.fromarray(a[:, :, 0] if a.ndim == 3 and a.shape[2] == 1 else a).save(path)


def get_files_pattern(d, pattern):
    """library.py:95-102."""
    return sorted(fnmatch.filter(os.listdir(d), pattern))


def list_video_files_at_dir(d):
    """library.py:104-117: the first extension that matches anything wins."""
    for pattern in ('*tiff', '*tif', '*png', '*jpg', '*jpeg', '*raw'):
        paths = get_files_pattern(d, pattern)
        if len(paths) > 0:
            return [os.path.join(d, p) for p in paths]
    raise AssertionError("%s is empty!" % d)


def load_image(path, ftype=8):
    """library.py:119-131: image scaled to [0,1] by its bit depth, float32."""
    return np.asarray(iio_read(path), dtype=np.float32) / (2 ** float(ftype) - 1)


def pathdiff(a, b):
    """library.py:134-140."""
    assert a[:len(b)] == b, "b should be a subfolder/subfile of a"
    res = os.path.dirname(a[len(b):])
    return res[1:] if res[0] == '/' else res


def warpedimagefile(wfolder, fromCode, toCode):
    """library.py:142-143."""
    return os.path.join(wfolder, fromCode + '_' + toCode + '.tif')


def print_dict(val_losses, suffix="_valLoss", savefile=None):
    """library.py:21-30."""
    message = "[" + ", ".join('%s: %.3f' % (k + suffix, v) for k, v in val_losses.items()) + "]"
    print(message)
    if savefile is not None:
        with open(savefile, "a") as log_file:
            log_file.write('%s\n' % message)


def define_transforms(opt=None):
    """library.py:52-69: T = ToTensor (HWC ndarray -> CHW tensor, no scaling for float input) then
    2x - 1; iT = (x + 1)/2 back to an [H,W,C] ndarray."""
    def T(x):
        return 2. * torch.from_numpy(np.ascontiguousarray(np.asarray(x).transpose(2, 0, 1))) - 1.

    def iT(x):
        return ((x + 1.) / 2.).permute(1, 2, 0).numpy()
    return T, iT
