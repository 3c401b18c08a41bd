"""util/util.py:9-20 of the reference (psnr) on the HIP runtime."""
import torch

from ._ops import dev_index, ops_runtime


def psnr(input, target, max_val):
    if not torch.is_tensor(input) or not torch.is_tensor(target):
        raise TypeError(f"Expected 2 torch tensors but got {type(input)} and {type(target)}")
    if input.shape != target.shape:
        raise TypeError(f"Expected tensors of equal shapes, but got {input.shape} and {target.shape}")
    _, p = ops_runtime(dev_index(input)).psnr_l1(input.float(), target.float())
    # the kernel reports 10*log10(4/mse); rescale for another max_val
    import math
    return torch.tensor(p + 10.0 * math.log10(max_val * max_val / 4.0))


def tensor2im(input_image, imtype=None, clip=False, iT=None, force3chan=False):
    """util/util.py:23-49 of the reference: first item of a [N,C,H,W] tensor in [-1,1] -> [H,W,C] image in [0,255]
    (float32 unless `imtype` says otherwise): what `save_images` writes.  NumPy input passes through, anything that
    is neither a tensor nor an array is returned untouched.  Three fp32 operations on the host copy, as there."""
    import numpy as np
    if torch.is_tensor(input_image):
        item = input_image.detach()[0].cpu()
        if iT is not None:
            img = iT(item) * 255.0
        else:
            chw = item.float().numpy()
            chw = np.tile(chw, (3, 1, 1)) if chw.shape[0] == 1 else chw        # gray -> three equal channels
            img = (chw.transpose(1, 2, 0) + 1) / 2.0 * 255.0
    elif isinstance(input_image, np.ndarray):
        img = input_image
    else:
        return input_image
    img = np.clip(img, 0, 255) if clip else img
    if force3chan and img.shape[2] == 4:
        img = img[:, :, :3]
    return img.astype(np.float32 if imtype is None else imtype)


def save_image(image_numpy, image_path):
    """util/util.py:52-59."""
    from ..library import iio_write
    iio_write(image_numpy, image_path)


def mkdir(path):
    """util/util.py:91-98."""
    import os
    if not os.path.exists(path):
        os.makedirs(path)
