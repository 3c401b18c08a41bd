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
    """util/util.py:23-49: [1,C,H,W] tensor in [-1,1] -> float32 [H,W,C] image in [0,255] (what
    `save_images` writes).  The conversion is three fp32 operations on the host copy, as there."""
    import numpy as np
    imtype = np.float32 if imtype is None else imtype
    if not isinstance(input_image, np.ndarray):
        if not isinstance(input_image, torch.Tensor):
            return input_image
        image_tensor = input_image.data
        if iT is None:
            image_numpy = image_tensor[0].cpu().float().numpy()
            if image_numpy.shape[0] == 1:
                image_numpy = np.tile(image_numpy, (3, 1, 1))
            image_numpy = (np.transpose(image_numpy, (1, 2, 0)) + 1) / 2.0 * 255.0
        else:
            image_numpy = iT(image_tensor[0].cpu()) * 255.0
    else:
        image_numpy = input_image
    if clip:
        image_numpy = np.clip(image_numpy, 0, 255)
    if force3chan and np.shape(image_numpy)[2] == 4:
        image_numpy = image_numpy[:, :, :3]
    return image_numpy.astype(imtype)


def save_image(image_numpy, image_path):
    """util/util.py:52-59."""
    from ..library import iio_write
    iio_write(image_numpy, image_path)


def mkdir(path):
    """util/util.py:91-98."""
    import os
    if not os.path.exists(path):
        os.makedirs(path)
