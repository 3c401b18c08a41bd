"""util/Hamilton_Adam_demo.py of the reference on the HIP runtime."""
from __future__ import annotations

import torch

from ._ops import dev_index, ops_runtime


class HamiltonAdam:
    def __init__(self, pattern):
        if pattern != 'gbrg':
            raise NotImplementedError("rvdd HamiltonAdam: only the 'gbrg' pattern is built "
                                      "(the only one the reference instantiates, recurrent_model.py:99)")
        self.pattern = pattern

    def to(self, *a, **k):
        return self

    def __call__(self, x):
        return self.forward(x)

    def forward(self, x):
        """[B,4k,H,W] -> [B,3k,2H,2W] (util/Hamilton_Adam_demo.py:249-289)."""
        return ops_runtime(dev_index(x)).demosaic(x.float())

    def remosaick(self, x):
        """util/Hamilton_Adam_demo.py:237-246 (pure indexing, no arithmetic)."""
        B, _, H, W = x.size()
        y = torch.zeros(B, 4, H // 2, W // 2, dtype=x.dtype, device=x.device)
        y[:, 0] = x[:, 1, 0::2, 0::2]
        y[:, 1] = x[:, 2, 0::2, 1::2]
        y[:, 2] = x[:, 0, 1::2, 0::2]
        y[:, 3] = x[:, 1, 1::2, 1::2]
        return y
