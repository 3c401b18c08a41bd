"""A shared op-only runtime handle (demosaic / warp / upsample need no weights)."""
from __future__ import annotations

import torch

from ..runtime import RvddRuntime

_OPS = {}


def ops_runtime(device: int = 0) -> RvddRuntime:
    rt = _OPS.get(device)
    if rt is None:
        rt = RvddRuntime("convunet", 0, 1, 16, 16, device)
        _OPS[device] = rt
    return rt


def dev_index(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise RuntimeError("rvdd ops take GPU tensors only (no CPU path)")
    return t.device.index or 0
