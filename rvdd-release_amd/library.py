"""library.CPPbridge of the reference (library.py:143-175) on the HIP runtime: the TV-L1 optical flow
that the reference computes with its native libBridge.so (libBridge.cpp:44-163)."""
from __future__ import annotations

import numpy as np
import torch

from .util._ops import ops_runtime


class CPPbridge(object):
    """Same constructor and method as the reference's ctypes bridge; `libpath` is accepted and
    ignored (the flow runs in librvdd_hip.so on `device`)."""

    def __init__(self, libpath=None, device: int = 0):
        self.device = device
        self.rt = ops_runtime(device)

    def TVL1_flow(self, Im1, Im2):
        """Im1, Im2: [h,w,c] images (numpy or torch; c = 1 or 4).  4-channel raw frames are reduced
        by the channel mean as in library.py:165-167; 3-channel input must be gray-converted by the
        caller (the reference uses skimage.rgb2gray, not reproduced here).
        Returns the flow as a float32 numpy array [h,w,2] such that Im2(x + flow) ~ Im1(x)."""
        a = Im1 if torch.is_tensor(Im1) else torch.as_tensor(np.asarray(Im1))
        b = Im2 if torch.is_tensor(Im2) else torch.as_tensor(np.asarray(Im2))
        if a.shape != b.shape:
            raise AssertionError("Both images Im1 and Im2 are supposed to share same size")
        if a.dim() != 3 or a.shape[2] not in (1, 4):
            raise NotImplementedError("rvdd TVL1_flow: pass [h,w,1] gray or [h,w,4] packed raw images")
        dev = torch.device("cuda", self.device)
        g1 = a.to(dev, torch.float32).mean(dim=2).contiguous()
        g2 = b.to(dev, torch.float32).mean(dim=2).contiguous()
        flow = self.rt.tvl1flow(g1, g2)
        return flow.permute(1, 2, 0).contiguous().cpu().numpy()
