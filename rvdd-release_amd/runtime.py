"""Thin object wrapper over one ``rvdd_t`` handle.

PyTorch-ROCm is plumbing here: it owns device memory (``tensor.data_ptr()``
is what crosses the C ABI) and the HIP stream; all arithmetic is in
``librvdd_hip.so``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib

ARCH_BY_NAME = {
    "convunet": _lib.ARCH_CONVUNET,
    "convunet+feat": _lib.ARCH_CONVUNET_FEAT,
    "next": _lib.ARCH_CONVNEXT,
    "next+feat": _lib.ARCH_CONVNEXT_FEAT,
}


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk_dev(t: torch.Tensor, shape: Tuple[int, ...], name: str, device: Optional[int] = None) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (rvdd has no CPU path)")
    if device is not None and t.device.index != device:
        raise RuntimeError(f"{name} lives on cuda:{t.device.index} but this runtime drives cuda:{device}")
    if tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")
    return t if t.is_contiguous() else t.contiguous()


def _batch_strided(t: Optional[torch.Tensor], shape: Tuple[int, ...], name: str, device: int):
    """A [B,C,h,w] input of a step: dense inside a sequence, any stride from one sequence to the next (a channel
    slice of the reference's `n` / `flow` tensors is exactly that).  -> (tensor, batch stride in floats); copies
    only what is not laid out like that."""
    if t is None:
        return None, 0
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (rvdd has no CPU path)")
    if t.device.index != device:
        raise RuntimeError(f"{name} lives on cuda:{t.device.index} but this runtime drives cuda:{device}")
    if tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")
    C, h, w = shape[1:]
    if t.stride()[1:] == (h * w, w, 1) and t.stride(0) >= C * h * w:
        return t, t.stride(0)
    t = t.contiguous()
    return t, C * h * w


class RvddRuntime:
    """One handle = one device, one (arch, future, B, H, W) configuration."""

    def __init__(self, arch: str, future: int, batch: int, height: int, width: int, device: int = 0):
        self.lib = _lib.load()
        self.arch = arch
        self.future = int(future)
        self.B, self.H, self.W = int(batch), int(height), int(width)
        self.device = int(device)
        self.feat = arch.endswith("+feat")
        cfg = _lib.RvddCfg(ARCH_BY_NAME[arch], self.future, self.B, self.H, self.W, self.device)
        h = C.c_void_p()
        rc = self.lib.rvdd_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"rvdd_create failed ({rc}): {self.lib.rvdd_last_error(None).decode()}")
        self.h = h
        self._tdev = torch.device("cuda", self.device)

    # -- helpers ----------------------------------------------------------
    def _check(self, rc: int, what: str):
        if rc != 0 and not (getattr(self, "h", None) is not None and self.h.value):
            raise RuntimeError(f"{what}: this RvddRuntime is closed (evicted from the denoiser's cache of frame sizes, or "
                               "closed explicitly) -- fetch a live one with net.runtime_for(B, H, W)")
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.rvdd_last_error(self.h).decode()}")

    def _stream(self) -> int:
        return torch.cuda.current_stream(self._tdev).cuda_stream

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.rvdd_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights (BaseModel.load_networks, models/base_model.py:173-196) ---
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        for k, v in sd.items():
            t = v.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            self._check(self.lib.rvdd_set_weight(self.h, k.encode(), t.data_ptr(), shape, t.dim()),
                        f"rvdd_set_weight({k})")
        self._check(self.lib.rvdd_finalize_weights(self.h), "rvdd_finalize_weights")

    # -- hot path -----------------------------------------------------------
    def reset(self):
        self._check(self.lib.rvdd_reset(self.h), "rvdd_reset")

    def set_option(self, name: str, value: int):
        """Options of recurrentModel that change what a step does; "no_warp" = --no_warp (flows then unused)."""
        self._check(self.lib.rvdd_set_option(self.h, name.encode(), int(value)), "rvdd_set_option")
        if name == "no_warp":
            self.no_warp = bool(value)

    def step(self, raw_prev, raw_cur, raw_next, flow_prev, flow_next, out=None) -> torch.Tensor:
        B, H, W = self.B, self.H, self.W
        rs, fs = (B, 4, H // 2, W // 2), (B, 2, H // 2, W // 2)
        if getattr(self, "no_warp", False):
            flow_prev = flow_next = None
        elif flow_prev is None:
            raise RuntimeError("rvdd_step: flow_prev is required (no --no_warp option set on this runtime)")
        raws = [_batch_strided(t, rs, n, self.device) for t, n in ((raw_prev, "raw_prev"), (raw_cur, "raw_cur"), (raw_next, "raw_next"))]
        flows = [_batch_strided(t, fs, n, self.device) for t, n in ((flow_prev, "flow_prev"), (flow_next, "flow_next"))]
        if raws[1][0] is None:
            raise RuntimeError("rvdd_step: raw_cur is required")

        def common_stride(items, dense):
            strides = {st for t, st in items if t is not None}
            if len(strides) > 1:        # slices of different tensors: fall back to dense copies
                return [(None if t is None else t.contiguous(), dense) for t, _ in items], dense
            return items, (strides.pop() if strides else dense)
        raws, rstride = common_stride(raws, rs[1] * rs[2] * rs[3])
        flows, fstride = common_stride(flows, fs[1] * fs[2] * fs[3])
        if out is None:
            out = torch.empty(B, 3, H, W, dtype=torch.float32, device=self._tdev)
        else:
            _chk_dev(out, (B, 3, H, W), "out", self.device)
            assert out.is_contiguous()
        self._check(self.lib.rvdd_step_strided(self.h, _ptr(raws[0][0]), _ptr(raws[1][0]), _ptr(raws[2][0]),
                                               _ptr(flows[0][0]), _ptr(flows[1][0]), rstride, fstride, _ptr(out),
                                               self._stream()), "rvdd_step")
        return out

    def get_state(self, want_feat: bool = True):
        B, H, W = self.B, self.H, self.W
        den = torch.empty(B, 3, H, W, dtype=torch.float32, device=self._tdev)
        feat = torch.empty(B, 48, H, W, dtype=torch.float32, device=self._tdev) if (self.feat and want_feat) else None
        self._check(self.lib.rvdd_get_state(self.h, _ptr(den), _ptr(feat), self._stream()), "rvdd_get_state")
        return den, feat

    def set_state(self, lastden=None, lastfeat=None):
        B, H, W = self.B, self.H, self.W
        if lastden is not None:
            lastden = _chk_dev(lastden, (B, 3, H, W), "lastden", self.device)
        if lastfeat is not None:
            lastfeat = _chk_dev(lastfeat, (B, 48, H, W), "lastfeat", self.device)
        self._check(self.lib.rvdd_set_state(self.h, _ptr(lastden), _ptr(lastfeat), self._stream()),
                    "rvdd_set_state")

    def psnr_l1(self, den: torch.Tensor, gt: torch.Tensor) -> Tuple[float, float]:
        """-> (L1*100, PSNR) as compute_losses (models/recurrent_model.py:512-525)."""
        den = _chk_dev(den, den.shape, "den", self.device)
        gt = _chk_dev(gt, den.shape, "gt", self.device)
        out = (C.c_float * 2)()
        self._check(self.lib.rvdd_psnr_l1(self.h, _ptr(den), _ptr(gt), den.numel(), out, self._stream()),
                    "rvdd_psnr_l1")
        return float(out[0]), float(out[1])

    # -- single ops -----------------------------------------------------------
    def unet_forward(self, x, feat_in=None):
        B, H, W = self.B, self.H, self.W
        x = _chk_dev(x, (B, 3 * (2 + self.future), H, W), "x", self.device)
        if feat_in is not None:
            feat_in = _chk_dev(feat_in, (B, 48, H, W), "feat_in", self.device)
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=self._tdev)
        fo = torch.empty(B, 48, H, W, dtype=torch.float32, device=self._tdev) if self.feat else None
        self._check(self.lib.rvdd_unet_forward(self.h, _ptr(x), _ptr(feat_in), _ptr(out), _ptr(fo),
                                               self._stream()), "rvdd_unet_forward")
        return out, fo

    def demosaic(self, raw: torch.Tensor) -> torch.Tensor:
        n, c, h, w = raw.shape
        raw = _chk_dev(raw, raw.shape, "raw", self.device)
        assert c % 4 == 0
        k = n * (c // 4)
        out = torch.empty(n, 3 * (c // 4), 2 * h, 2 * w, dtype=torch.float32, device=raw.device)
        self._check(self.lib.rvdd_demosaic_ha(self.h, _ptr(raw), k, h, w, _ptr(out), self._stream()),
                    "rvdd_demosaic_ha")
        return out

    def warp(self, x: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
        n, c, H, W = x.shape
        x = _chk_dev(x, x.shape, "x", self.device)
        flow = _chk_dev(flow, (n, 2, H, W), "flow", self.device)
        y = torch.empty_like(x)
        self._check(self.lib.rvdd_warp_bicubic(self.h, _ptr(x), _ptr(flow), n, c, H, W, _ptr(y),
                                               self._stream()), "rvdd_warp_bicubic")
        return y

    def upsample_factor_2(self, t: torch.Tensor, multiply_by: float = 1.0) -> torch.Tensor:
        *rem, c, h, w = t.shape
        t = _chk_dev(t, t.shape, "t", self.device)
        n = 1
        for r in rem:
            n *= r
        out = torch.empty(*rem, c, 2 * h, 2 * w, dtype=torch.float32, device=t.device)
        self._check(self.lib.rvdd_upsample_factor_2(self.h, _ptr(t), n, c, h, w, float(multiply_by),
                                                    _ptr(out), self._stream()), "rvdd_upsample_factor_2")
        return out

    def tvl1flow(self, I0: torch.Tensor, I1: torch.Tensor, want_iterations: bool = False):
        """[ny,nx] x2 -> flow [2,ny,nx] (libBridge tvl1flow, libBridge.cpp:44)."""
        ny, nx = I0.shape
        I0 = _chk_dev(I0, (ny, nx), "I0", self.device)
        I1 = _chk_dev(I1, (ny, nx), "I1", self.device)
        u = torch.empty(2, ny, nx, dtype=torch.float32, device=I0.device)
        it = C.c_int32(0)
        self._check(self.lib.rvdd_tvl1flow(self.h, _ptr(I0), _ptr(I1), _ptr(u), nx, ny,
                                           C.byref(it) if want_iterations else None, self._stream()), "rvdd_tvl1flow")
        return (u, int(it.value)) if want_iterations else u

    def tvl1flow_batch(self, I0: torch.Tensor, I1: torch.Tensor, want_iterations: bool = False):
        """[n,ny,nx] x2 -> flows [n,2,ny,nx]: n independent pairs, two per cooperative launch."""
        n, ny, nx = I0.shape
        I0 = _chk_dev(I0, (n, ny, nx), "I0", self.device)
        I1 = _chk_dev(I1, (n, ny, nx), "I1", self.device)
        u = torch.empty(n, 2, ny, nx, dtype=torch.float32, device=I0.device)
        it = (C.c_int32 * max(n, 1))()
        self._check(self.lib.rvdd_tvl1flow_batch(self.h, _ptr(I0), _ptr(I1), _ptr(u), n, nx, ny,
                                                 it if want_iterations else None, self._stream()), "rvdd_tvl1flow_batch")
        return (u, list(it)[:n]) if want_iterations else u

    def ppipe(self, img: torch.Tensor, rgb_gain: float, red_gain: float, blue_gain: float, iso: int,
              bit_depth: int, layout: str = "nchw", want_float: bool = False):
        """dataset/fwd_ppipe.py:48-77,131-141.  img [n,3,H,W] ("nchw") or [n,H,W,3] ("hwc"), any strides.
        -> uint8 [n,H,W,3] (and the float32 sRGB x 255 image before rounding)."""
        if not img.is_cuda or img.dtype != torch.float32 or img.dim() != 4:
            raise RuntimeError("ppipe: img must be a 4-D float32 GPU tensor (rvdd has no CPU path)")
        if img.device.index != self.device:
            raise RuntimeError(f"ppipe: img lives on cuda:{img.device.index} but this runtime drives cuda:{self.device}")
        if layout == "nchw":
            n, c, H, W = img.shape
            sn, sc, sy, sx = img.stride()
        elif layout == "hwc":
            n, H, W, c = img.shape
            sn, sy, sx, sc = img.stride()
        else:
            raise ValueError("layout must be 'nchw' or 'hwc'")
        if c != 3:
            raise AssertionError("The data should have 3 channels.")              # fwd_ppipe.py:129
        u8 = torch.empty(n, H, W, 3, dtype=torch.uint8, device=img.device)
        f32 = torch.empty(n, H, W, 3, dtype=torch.float32, device=img.device) if want_float else None
        self._check(self.lib.rvdd_ppipe(self.h, _ptr(img), n, H, W, sn, sc, sy, sx, int(bit_depth), float(rgb_gain),
                                        float(red_gain), float(blue_gain), int(iso), _ptr(u8), _ptr(f32),
                                        self._stream()), "rvdd_ppipe")
        return (u8, f32) if want_float else u8

    def srgb_metrics(self, a: torch.Tensor, b: torch.Tensor):
        """dataset/fwd_ppipe.py:79-86 on uint8 [n,H,W,3] images -> (psnr[n], ssim[n]) Python floats."""
        if a.shape != b.shape:
            raise RuntimeError(f"srgb_metrics: shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
        if not (a.is_cuda and b.is_cuda) or a.dtype != torch.uint8 or b.dtype != torch.uint8 or a.dim() != 4 \
                or a.shape[3] != 3:
            raise RuntimeError("srgb_metrics: a, b must be uint8 GPU tensors [n,H,W,3]")
        if a.device.index != self.device or b.device.index != self.device:
            raise RuntimeError(f"srgb_metrics: a, b must live on cuda:{self.device}, the device this runtime drives")
        a, b = a.contiguous(), b.contiguous()
        n, H, W, _ = a.shape
        ps, ss = (C.c_double * n)(), (C.c_double * n)()
        self._check(self.lib.rvdd_srgb_metrics(self.h, _ptr(a), _ptr(b), n, H, W, ps, ss, self._stream()),
                    "rvdd_srgb_metrics")
        return list(ps), list(ss)

    # -- measurement ------------------------------------------------------------
    def profile_enable(self, on: bool):
        self._check(self.lib.rvdd_profile_enable(self.h, 1 if on else 0), "rvdd_profile_enable")

    def profile_select(self, kernel_class=None, stride: int = 1):
        self._check(self.lib.rvdd_profile_select(self.h, None if kernel_class is None else kernel_class.encode(),
                                                 stride), "rvdd_profile_select")

    def profile_read(self):
        out = []
        for i in range(self.lib.rvdd_profile_count(self.h)):
            name = C.create_string_buffer(128)
            n = C.c_int64()
            ms, fl, by = C.c_double(), C.c_double(), C.c_double()
            self._check(self.lib.rvdd_profile_read(self.h, i, name, 128, C.byref(n), C.byref(ms),
                                                   C.byref(fl), C.byref(by)), "rvdd_profile_read")
            out.append(dict(name=name.value.decode(), launches=n.value, ms=ms.value, flops=fl.value,
                            bytes=by.value))
        return out

    def debug_conv_bench(self, variant: int, level: int = 0, iters: int = 20) -> float:
        ms = C.c_float()
        self._check(self.lib.rvdd_debug_conv_bench(self.h, variant, level, iters, C.byref(ms), self._stream()),
                    "rvdd_debug_conv_bench")
        return float(ms.value)

    def timer_start(self):
        self._check(self.lib.rvdd_timer_start(self.h, self._stream()), "rvdd_timer_start")

    def timer_stop_ms(self) -> float:
        ms = C.c_float()
        self._check(self.lib.rvdd_timer_stop_ms(self.h, self._stream(), C.byref(ms)), "rvdd_timer_stop_ms")
        return float(ms.value)
