"""CPU oracle for the RVDD recurrent denoise+demosaic inference path.

TEST INFRASTRUCTURE ONLY.  This module is the *checker*: it may be imported by
``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of
``bench.py`` -- never by the product package (``rvdd-release_amd/``), which
must fail loudly when its HIP library is missing instead of falling back here.

It restates, on the CPU in fp32 with plain ``torch.nn.functional`` ops plus a
few explicit stencils, what the reference computes on the hot path.  Every
function cites the reference file:line it follows (paths are relative to the
upstream tree, centreborelli/RVDD-release).

Parity status: PINNED.  The reference ships no golden vectors or tests for
this path, so the pin is the reference itself, imported
in the build container by ``tools/make_golden.py``; that script writes the
small fixtures under ``tests/golden/`` which ``tests/test_oracle_golden.py``
checks this file against (max-abs tolerances written in that test).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------
# A2  Hamilton-Adams demosaic  (util/Hamilton_Adam_demo.py:249-289)
# --------------------------------------------------------------------------


def pack_in_one(x: Tensor) -> Tensor:
    """[B,4,h,w] packed GBRG planes -> [B,2h,2w] CFA image.

    util/Hamilton_Adam_demo.py:226-234: ch0 -> (even row, even col),
    ch1 -> (even, odd), ch2 -> (odd, even), ch3 -> (odd, odd).
    """
    B, _, h, w = x.shape
    y = torch.zeros(B, 2 * h, 2 * w, dtype=x.dtype)
    y[:, 0::2, 0::2] = x[:, 0]
    y[:, 0::2, 1::2] = x[:, 1]
    y[:, 1::2, 0::2] = x[:, 2]
    y[:, 1::2, 1::2] = x[:, 3]
    return y


def remosaick(x: Tensor) -> Tensor:
    """[B,3,H,W] RGB -> [B,4,H/2,W/2] GBRG planes (Hamilton_Adam_demo.py:237-246)."""
    B, _, H, W = x.shape
    y = torch.zeros(B, 4, H // 2, W // 2, dtype=x.dtype)
    y[:, 0] = x[:, 1, 0::2, 0::2]
    y[:, 1] = x[:, 2, 0::2, 1::2]
    y[:, 2] = x[:, 0, 1::2, 0::2]
    y[:, 3] = x[:, 1, 1::2, 1::2]
    return y


def _shift(p: Tensor, dy: int, dx: int, r: int) -> Tensor:
    """View of the replicate-padded plane `p` ([B,H+2r,W+2r]) shifted by (dy,dx)."""
    H = p.shape[-2] - 2 * r
    W = p.shape[-1] - 2 * r
    return p[:, r + dy:r + dy + H, r + dx:r + dx + W]


def _rpad(x: Tensor, r: int) -> Tensor:
    return F.pad(x[:, None], (r, r, r, r), mode="replicate")[:, 0]


def hamilton_adams(x: Tensor) -> Tensor:
    """Hamilton-Adams demosaic of packed GBRG raw frames.

    x: [B, 4k, h, w] -> [B, 3k, 2h, 2w]  (util/Hamilton_Adam_demo.py:249-289).

    Written as explicit per-pixel stencils in the evaluation order of the
    reference's fixed-weight convolutions (row-major over the kernel window,
    Hamilton_Adam_demo.py:41-120).  All stencil weights are powers of two, so
    products are exact and only the order of the additions matters; the hard
    ``sign`` selections (:138-139, :168-169) make that order observable.
    """
    B0, C, h, w = x.shape
    x = x.reshape(-1, 4, h, w)
    B = x.shape[0]
    H, W = 2 * h, 2 * w
    c = pack_in_one(x)                                   # :261

    yy = torch.arange(H)[:, None]
    xx = torch.arange(W)[None, :]
    ev_y, ev_x = (yy % 2 == 0), (xx % 2 == 0)
    one = torch.ones(H, W)
    zero = torch.zeros(H, W)
    # mosaic_bayer_mask('gbrg') (:201-224): G at (e,e),(o,o); B at (e,o); R at (o,e)
    m_g = torch.where((ev_y & ev_x) | (~ev_y & ~ev_x), one, zero)
    m_b = torch.where(ev_y & ~ev_x, one, zero)
    m_r = torch.where(~ev_y & ev_x, one, zero)
    # algo2_mask('gbrg') (:190-192): maskGb = (e,e), maskGr = (o,o)
    m_gb = torch.where(ev_y & ev_x, one, zero)
    m_gr = torch.where(~ev_y & ~ev_x, one, zero)

    # ---- algo1: green (:123-142); 5x5 stencils on the replicate-padded CFA
    p = _rpad(c, 2)
    s = lambda dy, dx: _shift(p, dy, dx, 2)
    Kh = 0.5 * s(0, -1) + 0.5 * s(0, 1)
    Kv = 0.5 * s(-1, 0) + 0.5 * s(1, 0)
    Dh = (s(0, -2) + (-2.0) * s(0, 0)) + s(0, 2)
    Dv = (s(-2, 0) + (-2.0) * s(0, 0)) + s(2, 0)
    Fh = s(0, -1) + (-1.0) * s(0, 1)
    Fv = s(-1, 0) + (-1.0) * s(1, 0)
    rawh = Kh - Dh / 4
    rawv = Kv - Dv / 4
    CLh = Fh.abs() + Dh.abs()
    CLv = Fv.abs() + Dv.abs()
    sg = torch.sign(CLh - CLv)
    green = (1 + sg) * rawv / 2 + (1 - sg) * rawh / 2
    green = green * (1 - m_g) + c * m_g                  # :140

    # ---- algo2: red / blue (:145-172); 3x3 stencils, replicate pad 1
    gp = _rpad(green, 1)
    g = lambda dy, dx: _shift(gp, dy, dx, 1)
    gDh = (0.25 * g(0, -1) + (-0.5) * g(0, 0)) + 0.25 * g(0, 1)
    gDv = (0.25 * g(-1, 0) + (-0.5) * g(0, 0)) + 0.25 * g(1, 0)
    gDp = (g(-1, -1) + (-2.0) * g(0, 0)) + g(1, 1)
    gDn = (g(-1, 1) + (-2.0) * g(0, 0)) + g(1, -1)

    def algo2(xc: Tensor, m_o: Tensor, mGr: Tensor, mGb: Tensor) -> Tensor:
        xp = _rpad(xc, 1)
        q = lambda dy, dx: _shift(xp, dy, dx, 1)
        Kh2 = 0.5 * q(0, -1) + 0.5 * q(0, 1)
        Kv2 = 0.5 * q(-1, 0) + 0.5 * q(1, 0)
        Kp2 = 0.5 * q(-1, -1) + 0.5 * q(1, 1)
        Kn2 = 0.5 * q(-1, 1) + 0.5 * q(1, -1)
        Fp2 = (-1.0) * q(-1, -1) + q(1, 1)
        Fn2 = (-1.0) * q(-1, 1) + q(1, -1)
        Ch = mGr * (Kh2 - gDh)
        Cv = mGb * (Kv2 - gDv)
        Cp = m_o * (Kp2 - gDp / 4)
        Cn = m_o * (Kn2 - gDn / 4)
        CLp = m_o * (Fp2.abs() + gDp.abs())
        CLn = m_o * (Fn2.abs() + gDn.abs())
        sl = torch.sign(CLp - CLn)
        ch = (1 + sl) * Cn / 2 + (1 - sl) * Cp / 2
        return (ch + Ch + Cv) + xc

    red = algo2(c * m_r, m_b, m_gr, m_gb)                # :280 (mode 1)
    blue = algo2(c * m_b, m_r, m_gb, m_gr)               # :281 (mode 2 swaps Gr/Gb)
    y = torch.stack((red, green, blue), 1)               # :284
    return y.reshape(B0, -1, H, W)


# --------------------------------------------------------------------------
# A3  flow upsample  (util/flow_utils.py:159-174)
# --------------------------------------------------------------------------


def upsample_factor_2(t: Tensor, multiply_by: float = 1.0) -> Tensor:
    """[...,C,h,w] -> [...,C,2h,2w], bilinear, align_corners=True, times multiply_by."""
    *rem, C, h, w = t.shape
    up = F.interpolate(t.reshape(-1, C, h, w), scale_factor=2, mode="bilinear",
                       align_corners=True).reshape(*rem, C, 2 * h, 2 * w)
    return up * multiply_by


def upsample_factor_2_explicit(t: Tensor, multiply_by: float = 1.0) -> Tensor:
    """Closed form of :func:`upsample_factor_2` in the arithmetic the HIP warp
    kernel uses: src = dst * ((in-1)/(out-1)) in fp32, i0 = floor, lerp
    ``l0h*(l0w*v00 + l1w*v01) + l1h*(l0w*v10 + l1w*v11)``."""
    *rem, C, h, w = t.shape
    H, W = 2 * h, 2 * w
    x = t.reshape(-1, C, h, w)
    sy = torch.tensor((h - 1) / (H - 1) if H > 1 else 0.0, dtype=torch.float32)
    sx = torch.tensor((w - 1) / (W - 1) if W > 1 else 0.0, dtype=torch.float32)
    fy = torch.arange(H, dtype=torch.float32) * sy
    fx = torch.arange(W, dtype=torch.float32) * sx
    y0 = fy.floor().long().clamp(max=h - 1)
    x0 = fx.floor().long().clamp(max=w - 1)
    y1 = (y0 + 1).clamp(max=h - 1)
    x1 = (x0 + 1).clamp(max=w - 1)
    ly1 = (fy - y0.float())[:, None]
    lx1 = (fx - x0.float())[None, :]
    ly0, lx0 = 1.0 - ly1, 1.0 - lx1
    v00 = x[:, :, y0][:, :, :, x0]
    v01 = x[:, :, y0][:, :, :, x1]
    v10 = x[:, :, y1][:, :, :, x0]
    v11 = x[:, :, y1][:, :, :, x1]
    up = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11)
    return (up * multiply_by).reshape(*rem, C, H, W)


# --------------------------------------------------------------------------
# A5  bicubic backward warp  (util/flow_utils.py:70-102)
# --------------------------------------------------------------------------


def warp(x: Tensor, flow: Tensor) -> Tensor:
    """Backward warp with ``grid_sample(bicubic, border, align_corners=True)``.

    x [B,C,H,W], flow [B,2,H,W] (x-displacement first) -> [B,C,H,W].
    Follows util/flow_utils.py:83-99; the validity mask (:95-96) is discarded
    by every hot-path caller (models/recurrent_model.py:151,154,297).
    """
    B, C, H, W = x.shape
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    grid = torch.stack((xx, yy), 0)[None].float()
    vgrid = grid + flow
    gx = 2.0 * vgrid[:, 0] / (W - 1) - 1.0
    gy = 2.0 * vgrid[:, 1] / (H - 1) - 1.0
    g = torch.stack((gx, gy), -1)
    return F.grid_sample(x, g, padding_mode="border", mode="bicubic", align_corners=True)


def _cubic_w(t: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Cubic-convolution weights, A = -0.75 (ATen UpSample.h get_cubic_upsample_coefficients)."""
    A = -0.75
    def c1(x):  # |x| <= 1
        return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0
    def c2(x):  # 1 < |x| < 2
        return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A
    return c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)


def warp_explicit(x: Tensor, flow: Tensor) -> Tensor:
    """Explicit 16-tap statement of :func:`warp` -- the arithmetic the HIP warp
    kernel implements: fp32 round trip through the [-1,1] normalisation,
    ``ix = ((g+1)/2)*(W-1)``, x0 = floor(ix), taps x0-1..x0+2 each clamped to
    [0,W-1] individually, separable A=-0.75 weights, x first then y."""
    B, C, H, W = x.shape
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    vx = xx[None].float() + flow[:, 0]
    vy = yy[None].float() + flow[:, 1]
    gx = 2.0 * vx / (W - 1) - 1.0
    gy = 2.0 * vy / (H - 1) - 1.0
    ix = ((gx + 1.0) / 2.0) * (W - 1)
    iy = ((gy + 1.0) / 2.0) * (H - 1)
    x0 = ix.floor()
    y0 = iy.floor()
    wx = _cubic_w(ix - x0)
    wy = _cubic_w(iy - y0)
    x0 = x0.long()
    y0 = y0.long()
    out = torch.zeros_like(x)
    bidx = torch.arange(B)[:, None, None]
    for j in range(4):
        yc = (y0 - 1 + j).clamp(0, H - 1)
        row = torch.zeros_like(x)
        for i in range(4):
            xc = (x0 - 1 + i).clamp(0, W - 1)
            v = x[bidx, :, yc, xc].permute(0, 3, 1, 2)     # [B,C,H,W]
            row = row + v * wx[i][:, None]
        out = out + row * wy[j][:, None]
    return out


# --------------------------------------------------------------------------
# A6-A8  convunet family  (networks/unet.py:544-588, 595-825)
# --------------------------------------------------------------------------


def zero_pad_features(size: Sequence[int], x: Tensor) -> Tensor:
    """Centred zero-pad of x up to `size` (networks/unet.py:151-170)."""
    if tuple(size) == tuple(x.shape):
        return x
    tmp = torch.zeros(tuple(size), dtype=x.dtype)
    sy = int((tmp.shape[2] - x.shape[2]) / 2)
    sx = int((tmp.shape[3] - x.shape[3]) / 2)
    tmp[:, :, sy:sy + x.shape[2], sx:sx + x.shape[3]] = x
    return tmp


def convunet_forward(sd: Dict[str, Tensor], x: Tensor,
                     old_features: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """``convunet-mode=fixedfeatures[+feat]`` forward.

    sd: the checkpoint state_dict (keys as in SURVEY.md section 8a/A12).
    x [B,Cin,H,W]; old_features [B,48,H,W] for the +feat net, else None.
    Returns (out [B,3,H,W], new_features or None).
    networks/unet.py:544-588 (UNet.forward) as specialised at :595-720 and,
    for +feat, :725-825 (preprocessing conv without activation :742, concat
    :743, hook on PostConvs[-2] = post-ReLU map :808-812).
    """
    def conv(name: str, t: Tensor, relu: bool) -> Tensor:
        t = F.conv2d(t, sd[name + ".weight"], sd[name + ".bias"], padding=1)
        return F.relu(t) if relu else t

    feat = "preprocessing_layer.weight" in sd
    if feat:
        if old_features is None:
            raise Exception("Old features is None, please call get_rec_nil_features first.")
        y = conv("preprocessing_layer", x, relu=False)
        x = torch.cat([y, old_features], 1)

    skips: List[Tensor] = []
    for i in range(4):
        x = conv(f"EncoderConvs.{i}.blocks.0.0", x, True)
        x = conv(f"EncoderConvs.{i}.blocks.1.0", x, True)
        skips.append(x)
        if i < 3:
            x = F.max_pool2d(conv(f"EncoderDown.{i}.conv", x, False), 2)   # :207-208

    d = skips[-1]
    s = d
    for i in range(2):                                                      # :561-567
        d = conv(f"bottleneck.{i}.0", d, True)
        s = s + d
    d = s

    for i in range(3):                                                      # :570-579
        d = F.interpolate(d, scale_factor=2, mode="bilinear")              # align_corners=False
        d = conv(f"DecoderUp.{i}.up.1", d, True)
        e = skips[-(i + 2)]
        d = zero_pad_features(e.shape, d)
        d = torch.cat((e, d), 1)                                            # (encoder, decoder) :541
        d = conv(f"DecoderConvs.{i}.blocks.0.0", d, True)
        d = conv(f"DecoderConvs.{i}.blocks.1.0", d, True)

    f = conv("PostConvs.0.0", d, True)
    out = F.conv2d(f, sd["PostConvs.1.weight"], sd["PostConvs.1.bias"])
    return out, (f if feat else None)


# --------------------------------------------------------------------------
# A9-A10  ConvNeXtUnet family  (networks/new_unet.py)
# --------------------------------------------------------------------------


def layer_norm_c(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    """Per-pixel channel LayerNorm, biased variance (networks/new_unet.py:23-28)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[None, :, None, None] * x + b[None, :, None, None]


def convnext_block(sd: Dict[str, Tensor], pre: str, x: Tensor) -> Tensor:
    """One ConvBlock (networks/new_unet.py:74-103)."""
    if pre + ".proj.weight" in sd:
        x = F.conv2d(x, sd[pre + ".proj.weight"], sd[pre + ".proj.bias"])
    r = F.conv2d(x, sd[pre + ".block.0.weight"], sd[pre + ".block.0.bias"], padding=3,
                 groups=x.shape[1])
    r = layer_norm_c(r, sd[pre + ".block.1.weight"], sd[pre + ".block.1.bias"])
    r = F.conv2d(r, sd[pre + ".block.2.weight"], sd[pre + ".block.2.bias"])
    r = F.gelu(r)                                                          # exact erf
    r = F.conv2d(r, sd[pre + ".block.4.weight"], sd[pre + ".block.4.bias"])
    return x + sd[pre + ".layerscale.layerscale"][None, :, None, None] * r


def convnext_forward(sd: Dict[str, Tensor], x: Tensor,
                     old_features: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """``newunet[-mode=feat]`` forward (networks/new_unet.py:332-362, 365-430)."""
    feat = any(k.startswith("preprocessing_layer.") for k in sd)
    if feat:
        if old_features is None:
            raise Exception("Old features is None, please call get_rec_nil_features first.")
        y = convnext_block(sd, "preprocessing_layer.blocks.0", x)
        x = torch.cat([y, old_features], 1)

    mem: List[Tensor] = []
    for i in range(4):
        x = convnext_block(sd, f"encoder_convs.{i}.blocks.0", x)
        x = convnext_block(sd, f"encoder_convs.{i}.blocks.1", x)
        mem.append(x)
        if i < 3:
            x = convnext_block(sd, f"encoder_downs.{i}.postconv", F.max_pool2d(x, 2))  # :200-204
    x = convnext_block(sd, "bottleneck.blocks.0", x)
    x = convnext_block(sd, "bottleneck.blocks.1", x)
    for i in range(3):
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)      # :145-149
        x = convnext_block(sd, f"decoder_ups.{i}.postconv", x)
        e = mem[-(i + 2)]
        x = zero_pad_features(e.shape, x)
        x = torch.cat((x, e), 1)                                           # (decoder, encoder) :326
        x = convnext_block(sd, f"decoder_convs.{i}.blocks.0", x)
        x = convnext_block(sd, f"decoder_convs.{i}.blocks.1", x)
    x = convnext_block(sd, "postprocessing.0.blocks.0", x)
    f = convnext_block(sd, "postprocessing.0.blocks.1", x)                 # hooked :414-417
    out = F.conv2d(f, sd["postprocessing.1.weight"], sd["postprocessing.1.bias"])
    return out, (f if feat else None)


def net_forward(sd, x, old_features=None):
    if any(k.startswith("encoder_convs.") for k in sd):
        return convnext_forward(sd, x, old_features)
    return convunet_forward(sd, x, old_features)


def net_has_feat(sd) -> bool:
    return any(k.startswith("preprocessing_layer.") for k in sd)


def net_input_nc(sd) -> int:
    for k in ("preprocessing_layer.weight", "EncoderConvs.0.blocks.0.0.weight",
              "preprocessing_layer.blocks.0.proj.weight", "encoder_convs.0.blocks.0.proj.weight"):
        if k in sd:
            return int(sd[k].shape[1])
    raise KeyError("cannot infer input channels")


# --------------------------------------------------------------------------
# A11  losses  (models/recurrent_model.py:512-525, util/util.py:9-20)
# --------------------------------------------------------------------------


def psnr(a: Tensor, b: Tensor, max_val: float = 2.0) -> float:
    mse = F.mse_loss(a, b, reduction="mean")
    return float(10 * torch.log10(max_val * max_val / mse))


def l1_loss(a: Tensor, b: Tensor, lambda_l1: float = 100.0) -> float:
    return float(F.l1_loss(a, b) * lambda_l1)


# --------------------------------------------------------------------------
# A1 + A4  the recurrence state machine (models/recurrent_model.py:105-349)
# --------------------------------------------------------------------------


class RecurrentOracle:
    """Streaming restatement of ``recurrentModel.set_input`` + ``forward`` (test
    branch: TD=1, D=1, unrollings=1; models/recurrent_model.py:161-349).

    One :meth:`step` = one output frame.  Inputs are what the reference's
    dataloader hands over (data/infer4rec_dataset.py:158-171): packed raw
    frames in [-1,1] at raw resolution and raw-resolution flows.
    """

    def __init__(self, sd: Dict[str, Tensor], future: int = 0, threads: Optional[int] = None, no_warp: bool = False,
                 prev_noisy_frame: bool = False, warp_raw: bool = False):
        self.warp_raw = bool(warp_raw)                                      # --warp_raw (:149-152, :128: the flow is NOT upsampled)
        self.prev_noisy_frame = bool(prev_noisy_frame)                      # --prev_noisy_frame (:335-337)
        self.no_warp = bool(no_warp)                                        # --no_warp: warp_frame returns its input (:137-159)
        self.sd = {k: v.float() for k, v in sd.items()}
        self.feat = net_has_feat(self.sd)
        self.future = int(future)
        cin = net_input_nc(self.sd)
        if cin != 3 * (2 + self.future):
            raise ValueError(f"checkpoint expects {cin} input channels, future={future}")
        self.lastden: Optional[Tensor] = None
        self.lastfeat: Optional[Tensor] = None
        if threads:
            torch.set_num_threads(threads)

    @torch.no_grad()
    def step(self, raw_prev: Tensor, raw_cur: Tensor, raw_next: Optional[Tensor],
             flow_prev: Tensor, flow_next: Optional[Tensor], first: bool) -> Tensor:
        """raw_* [B,4,h,w]; flow_* [B,2,h,w] (cur->prev, cur->next). -> [B,3,2h,2w]."""
        n_cur = hamilton_adams(raw_cur)                                    # :125-126
        fl_prev = None if self.no_warp else (flow_prev if self.warp_raw else upsample_factor_2(flow_prev, multiply_by=2))   # :128-129
        B, _, H, W = n_cur.shape
        if first or self.lastden is None:                                   # :233-245
            self.lastden = hamilton_adams(raw_prev)
            if self.feat:
                self.lastfeat = torch.zeros(B, 48, H, W)
        if self.no_warp:
            warped = self.lastden
        elif self.warp_raw:
            warped = hamilton_adams(warp(remosaick(self.lastden), fl_prev))
        else:
            warped = warp(self.lastden, fl_prev)                           # :281-287
        feat_in = None
        if self.feat:
            feat_in = self.lastfeat if self.no_warp else warp(self.lastfeat, fl_prev)     # :290-297
        parts = [warped, n_cur]                                            # :299-311
        if self.future:
            n_next = hamilton_adams(raw_next)
            if self.no_warp:
                parts.append(n_next)
            elif self.warp_raw:
                parts.append(hamilton_adams(warp(remosaick(n_next), flow_next)))
            else:
                fl_next = upsample_factor_2(flow_next, multiply_by=2)
                parts.append(warp(n_next, fl_next))                        # :314-324
        netinput = torch.cat(parts, 1)
        den, f = net_forward(self.sd, netinput, feat_in)                   # :327
        self.lastden = n_cur.clone() if self.prev_noisy_frame else den.clone()   # :335-337
        if self.feat:
            self.lastfeat = f                                              # :339-345
        return den

    def run_sequence(self, raw: Tensor, flow_prev: Tensor, flow_next: Optional[Tensor] = None
                     ) -> Tensor:
        """raw [T,4,h,w]; flow_prev[t] = flow t->t-1, flow_next[t] = flow t->t+1
        ([T,2,h,w]; entry 0 of flow_prev / last of flow_next unused).
        Returns the T-1-future outputs stacked [T-1-f,3,H,W] (B=1)."""
        T = raw.shape[0]
        outs = []
        for t in range(1, T - self.future):
            rn = raw[t + 1][None] if self.future else None
            fn = flow_next[t][None] if self.future else None
            outs.append(self.step(raw[t - 1][None], raw[t][None], rn, flow_prev[t][None], fn,
                                  first=(t == 1))[0])
        return torch.stack(outs, 0)
