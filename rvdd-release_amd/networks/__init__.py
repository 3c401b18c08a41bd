"""``--netDenoiser`` registry for the HIP runtime.

Mirrors networks/__init__.py:121-198 of the reference (``define_net_arch``,
``parse_kwargs``) for the two families that have shipped checkpoints; the
objects returned keep the reference nets' calling surface
(``net(x)``, ``set_rec_features``, ``get_current_features``,
``get_rec_nil_features``, ``NoPF``, ``load_state_dict``, ``state_dict``,
``eval``) but run on ``librvdd_hip.so``.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional

import torch

from ..runtime import RvddRuntime


def parse_kwargs(netG: str) -> dict:
    """``name-k=v-k=v`` mini-language (networks/__init__.py:179-198)."""
    keys = netG.split('-')[1:]
    kwargs = {}
    for item in keys:
        k, v = item.split('=')
        if v.isnumeric():
            v = int(v)
        elif v.lower() == "none":
            v = None
        elif v.lower() in ("true", "false", "yes", "no", "on", "off", "y", "n", "t", "f"):
            v = v.lower() in ("true", "yes", "on", "y", "t")
        kwargs[k] = v
    return kwargs


class HipDenoiser:
    """A denoiser network living in ``librvdd_hip.so``.

    One runtime handle per (B, H, W) seen, created lazily like the reference
    nets are size-agnostic; the weights are kept on the host to (re)upload.
    """

    _arch = "convunet"
    _name = None          # the reference's class-level lookup attribute (unet.py:14-24)

    def __init__(self, in_channels: int, out_channels: int = 3, device: int = 0, **kwargs):
        if out_channels != 3:
            raise NotImplementedError("rvdd: only output_nc=3 (joint denoise+demosaic) is built")
        if in_channels not in (6, 9):
            raise NotImplementedError(f"rvdd: input channels must be 6 or 9, got {in_channels}")
        unsupported = {k: v for k, v in kwargs.items() if k not in ("depth",) or (k == "depth" and v != 4)}
        if unsupported:
            raise NotImplementedError(f"rvdd: netDenoiser options {unsupported} are not built "
                                      "(no shipped checkpoint uses them)")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.future = in_channels // 3 - 2
        self.device_index = device
        self.filters = 48
        self._sd: Optional[Dict[str, torch.Tensor]] = None
        self._rt: "OrderedDict[tuple, RvddRuntime]" = OrderedDict()     # least recently used first
        self.old_features: Optional[torch.Tensor] = None
        self.training = False

    # -- torch.nn.Module look-alikes ---------------------------------------
    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("rvdd is an inference runtime; training is out of scope")
        return self

    def parameters(self):
        return iter(() if self._sd is None else self._sd.values())

    def state_dict(self):
        return OrderedDict() if self._sd is None else OrderedDict(self._sd)

    def load_state_dict(self, state_dict, strict: bool = True):
        """Strict by construction: the runtime rejects unknown / missing keys
        (the reference loads with strict=False, base_model.py:196)."""
        self._sd = OrderedDict((k, v.detach().to("cpu", torch.float32).contiguous())
                               for k, v in state_dict.items())
        for rt in self._rt.values():
            rt.close()
        self._rt.clear()

    def to(self, *a, **k):
        return self

    # -- runtime handles -------------------------------------------------------
    # A runtime owns its workspace (about 1.5 GB per 720p sequence): keep the handles of the last few frame
    # sizes only.  A validation run sees one or two sizes; a loop over many sizes must not pile them up.
    MAX_CACHED_RUNTIMES = 2

    def runtime_for(self, B: int, H: int, W: int, pin: bool = False) -> RvddRuntime:
        """The runtime of one frame size.  `pin=True` (recurrentModel.forward) marks it as the one that holds a
        video's recurrent state: it is never evicted by calls at other sizes, only replaced by the next pin."""
        key = (B, H, W)
        if pin:
            self._pinned = key
        rt = self._rt.get(key)
        if rt is not None:
            self._rt.move_to_end(key)
            return rt
        if self._sd is None:
            raise RuntimeError("rvdd: weights not loaded (call load_state_dict / model.setup first)")
        evictable = [k for k in self._rt if k != getattr(self, "_pinned", None)]      # least recently used first
        while len(self._rt) >= self.MAX_CACHED_RUNTIMES and evictable:
            self._rt.pop(evictable.pop(0)).close()
        rt = RvddRuntime(self._arch, self.future, B, H, W, self.device_index)
        rt.load_state_dict(self._sd)
        self._rt[key] = rt
        return rt

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        B, C, H, W = x.shape
        rt = self.runtime_for(B, H, W)
        out, _ = rt.unet_forward(x, None)
        return out

    forward = __call__


class HipDenoiserFeat(HipDenoiser):
    """+feat variants: recurrent 48-channel features (networks/unet.py:725-825,
    networks/new_unet.py:365-430)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.NoPF = 1

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if self.old_features is None:
            raise Exception('Old features is None, please call get_rec_nil_features first.')
        B, C, H, W = x.shape
        rt = self.runtime_for(B, H, W)
        out, feat = rt.unet_forward(x, self.old_features)
        self.old_features = feat                       # the forward hook (unet.py:811-812)
        return out

    forward = __call__

    def set_rec_features(self, features: List[torch.Tensor]):
        self.old_features = features[0]

    def get_current_features(self):
        return [self.old_features]

    def get_rec_nil_features(self, B, H, W, device=None, non_blocking=None):
        self.old_features = torch.zeros(B, self.filters, H, W, dtype=torch.float32,
                                        device=torch.device("cuda", self.device_index))
        return [self.old_features]


class UNet_FixedFeatures(HipDenoiser):
    _arch, _name = "convunet", "fixedfeatures"


class UNet_FixedFeatures_feat(HipDenoiserFeat):
    _arch, _name = "convunet+feat", "fixedfeatures+feat"


class NewUNet(HipDenoiser):
    _arch = "next"


class NewUNet_feat(HipDenoiserFeat):
    _arch, _name = "next+feat", "feat"


def get_UNet_cls(mode: str):
    for cls in (UNet_FixedFeatures, UNet_FixedFeatures_feat):
        if cls._name == mode.lower():
            return cls
    raise Exception(f'Provided mode "{mode}" does not exist.')


def define_net_arch(input_nc, output_nc, netG, init_type='normal', init_gain=0.02, gpu_ids=[], NoPF=-1):
    """String -> network, as networks/__init__.py:121-176.  ``init_type`` /
    ``init_gain`` are accepted and ignored (weights come from a checkpoint);
    the net is placed on ``gpu_ids[0]`` and NOT wrapped in DataParallel
    (the wrapper is a no-op at validation batch size and breaks feature
    recurrence)."""
    dev = gpu_ids[0] if gpu_ids else 0
    if "newunet" in netG:
        kwargs = parse_kwargs(netG)
        cls = NewUNet
        if kwargs.get('mode') == 'feat':
            cls = NewUNet_feat
        kwargs.pop('mode', None)
        return cls(input_nc, output_nc, device=dev, **kwargs)
    elif "convunet" in netG:
        kwargs = parse_kwargs(netG)
        mode = kwargs.pop('mode', 'default')
        if mode in ("default", "concat"):
            raise NotImplementedError("rvdd: plain `convunet` (growing filters) has no shipped checkpoint; "
                                      "use convunet-mode=fixedfeatures[+feat]")
        cls = get_UNet_cls(mode)
        return cls(input_nc, output_nc, device=dev, **kwargs)
    raise NotImplementedError(f"Generator model name {netG} is not recognized")
