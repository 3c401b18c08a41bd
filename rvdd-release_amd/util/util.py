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
