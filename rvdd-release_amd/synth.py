"""Deterministic synthetic raw video sequences for tests and ``bench.py``.

Follows the recipe of SURVEY.md section 8d: a smooth multi-frequency colour
texture in linear RGB, translated by a known sub-pixel velocity per sequence
(so the optical flow is analytic instead of TV-L1), mapped to the 12-bit DN
range of the reference's data synthesis, GBRG-mosaicked, corrupted by the
reference's heteroscedastic Gaussian noise model and normalised the way the
reference's loader does.

Reference formulas restated here (dataset synthesis is out of scope as code):
  * noise: ``sigma^2 = a*u - b`` in 12-bit DN, clipped at 0, with (a,b) =
    (8.0034, 2043.51144) for ISO3200 and (28.3015, 6307.62081) for ISO12800
    (dataset/generate_raw_from_RGB.py:186-189);
  * DN range [266,3610] (ISO3200) / [268,4075] (ISO12800) (:170-179);
  * GBRG packing ch0=G(even,even) ch1=B(even,odd) ch2=R(odd,even)
    ch3=G(odd,odd) (:86-96, util/Hamilton_Adam_demo.py:226-234);
  * loader normalisation ``/4095`` then ``2x-1`` (library.py:117-129, 63-65).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch

ISO_PARAMS = {
    3200: dict(a=8.0034, b=2043.51144, lo=266.0, hi=3610.0),
    12800: dict(a=28.3015, b=6307.62081, lo=268.0, hi=4075.0),
}


CAM_GAINS = (0.5, 1.0, 0.6)   # approx. 1/red_gain, 1, 1/blue_gain


@dataclass
class SynthSequence:
    raw: torch.Tensor        # [T,4,h,w]  packed noisy GBRG raw in [-1,1]
    flow_prev: torch.Tensor  # [T,2,h,w]  raw-res flow t -> t-1 (entry 0 unused)
    flow_next: torch.Tensor  # [T,2,h,w]  raw-res flow t -> t+1 (last entry unused)
    gt: torch.Tensor         # [T,3,H,W]  clean linear RGB in [-1,1]


def _texture(xs: torch.Tensor, ys: torch.Tensor, gen: torch.Generator, nfreq: int = 6
             ) -> torch.Tensor:
    """Smooth colour texture in [0,1]; xs, ys [H,W] pixel coordinates -> [3,H,W]."""
    dev = xs.device
    out = []
    for _ in range(3):
        f = torch.rand(nfreq, 2, generator=gen, device="cpu") * 0.12 + 0.004
        sign = torch.where(torch.rand(nfreq, 2, generator=gen) < 0.5, -1.0, 1.0)
        f = (f * sign).to(dev)
        ph = (torch.rand(nfreq, generator=gen) * 2 * math.pi).to(dev)
        amp = (torch.rand(nfreq, generator=gen) * 0.8 + 0.2).to(dev)
        acc = torch.zeros_like(xs)
        for k in range(nfreq):
            acc = acc + amp[k] * torch.sin(2 * math.pi * (f[k, 0] * xs + f[k, 1] * ys) + ph[k])
        out.append(0.5 + 0.5 * acc / amp.sum())
    tex = torch.stack(out, 0)
    # Scene-referred look: the reference's data are "unprocessed" sRGB frames
    # (inverse tone curve, gamma expansion, inverse white balance; dataset/
    # generate_raw_from_RGB.py:99-127), i.e. dark, with R and B attenuated
    # relative to G.  A cubic + camera-space gains reproduces those
    # statistics; on a flat bright texture the shipped checkpoints are out of
    # distribution and the recurrence drifts.
    gains = torch.tensor(CAM_GAINS, device=dev)[:, None, None]
    return tex.pow(3.0) * 0.4 * gains


def make_sequence(T: int, H: int, W: int, iso: int = 3200, seed: int = 0,
                  device: str = "cpu", max_speed: float = 3.0) -> SynthSequence:
    """One synthetic sequence of T frames at RGB size HxW (both even)."""
    assert H % 2 == 0 and W % 2 == 0
    P = ISO_PARAMS[iso]
    gen = torch.Generator(device="cpu").manual_seed(int(seed))
    dev = torch.device(device)
    h, w = H // 2, W // 2
    ang = float(torch.rand(1, generator=gen)) * 2 * math.pi
    spd = float(torch.rand(1, generator=gen)) * (max_speed - 0.5) + 0.5
    vx, vy = spd * math.cos(ang), spd * math.sin(ang)

    yy, xx = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32),
                            torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
    tex_state = gen.get_state()
    frames = []
    for t in range(T):
        gen.set_state(tex_state)                       # same texture, translated
        frames.append(_texture(xx - vx * t, yy - vy * t, gen))
    clean = torch.stack(frames, 0)                      # [T,3,H,W] in [0,1]
    dn = P["lo"] + clean * (P["hi"] - P["lo"])          # 12-bit DN

    # GBRG mosaic of the clean DN image, then noise
    g0 = dn[:, 1, 0::2, 0::2]
    b = dn[:, 2, 0::2, 1::2]
    r = dn[:, 0, 1::2, 0::2]
    g1 = dn[:, 1, 1::2, 1::2]
    u = torch.stack((g0, b, r, g1), 1)                  # [T,4,h,w]
    ngen = torch.Generator(device="cpu").manual_seed(int(seed) * 7919 + 13)
    z = torch.randn(u.shape, generator=ngen).to(dev)
    noisy = u + torch.sqrt(torch.clamp(P["a"] * u - P["b"], min=0.0)) * z
    raw = 2.0 * (noisy / 4095.0) - 1.0
    gt = 2.0 * (dn / 4095.0) - 1.0

    # analytic flows at raw resolution (+ a small smooth perturbation so that
    # bicubic taps are not degenerate)
    yr, xr = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32),
                            torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
    fl_p, fl_n = [], []
    for t in range(T):
        px = 0.1 * torch.sin(0.05 * xr + 0.03 * yr + 0.7 * t)
        py = 0.1 * torch.cos(0.04 * xr - 0.06 * yr + 0.3 * t)
        fl_p.append(torch.stack((-vx / 2 + px, -vy / 2 + py), 0))
        fl_n.append(torch.stack((vx / 2 - px, vy / 2 - py), 0))
    return SynthSequence(raw=raw.contiguous(), flow_prev=torch.stack(fl_p, 0).contiguous(),
                         flow_next=torch.stack(fl_n, 0).contiguous(), gt=gt.contiguous())
