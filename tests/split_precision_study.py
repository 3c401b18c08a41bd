#!/usr/bin/env python3
"""How exact is a dense conv whose f32 operands are split into narrow floats and multiplied on the f16 / bf16 matrix
pipe (16x the f32 MFMA rate on gfx950) with f32 accumulation?  CPU study on the oracle (a test helper, not the product):

    python tests/split_precision_study.py [fixture ...]

Every dense conv of the oracle (F.conv2d with groups == 1: the 3x3 convs of convunet, the 1x1 convs of ConvNeXt's MLP)
is replaced by the sum of the partial products a split scheme keeps, each computed in float64 (so what is measured is
the scheme's own error, not the accumulation order), and the whole 30 / 90-frame recurrence of the long fixtures is run
against the REFERENCE's stored frames and per-frame PSNR.  Schemes:

    f32        the oracle as it is (the floor: the oracle's own distance to the reference)
    f16x2p3    x = hi + lo in f16 (round-to-zero hi), products hi.hi + hi.lo + lo.hi
    f16x2p4    the same and lo.lo
    bf16x3p6   x = h + m + l in bf16, products of order <= 2 (hh hm mh mm hl lh)
    bf16x3p3   hh hm mh only (16 bits)
    bf16x2p3   x = h + l in bf16, hh hl lh
    wino-...   the 3x3 convs as Winograd F(2x2,3x3) with the element-wise products on split operands (the transformed
               patches and filters are what is split)

Also prints the largest |operand| any dense conv saw (f16 overflows at 65504).
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import LONG, load_long, load_weights  # noqa: E402
import rvdd_oracle as O  # noqa: E402

REAL = F.conv2d
STATS = {"max_x": 0.0, "max_w": 0.0}


def rtz_f16(x):
    """f32 -> f16 toward zero, clamped to the largest finite f16 (what v_cvt_pkrtz_f16_f32 does)."""
    h = x.to(torch.float16)
    hf = h.float()
    over = hf.abs() > x.abs()
    nxt = torch.nextafter(h, torch.zeros_like(h))
    h = torch.where(over, nxt, h)
    big = torch.tensor(65504.0, dtype=torch.float16)
    return torch.where(torch.isinf(h), torch.sign(x).to(torch.float16) * big, h)


def split_f16(x):
    hi = rtz_f16(x)
    lo = (x - hi.float()).to(torch.float16)          # exact difference, rounded to nearest (subnormals kept)
    return [hi.double(), lo.double()]


def split_bf16(x, n):
    parts, r = [], x.clone()
    for _ in range(n):
        p = r.to(torch.bfloat16)
        parts.append(p.double())
        r = r - p.float()
    return parts


SCHEMES = {
    "f16x2p3": (split_f16, [(0, 0), (0, 1), (1, 0)]),
    "f16x2p4": (split_f16, [(0, 0), (0, 1), (1, 0), (1, 1)]),
    "bf16x3p6": (lambda t: split_bf16(t, 3), [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]),
    "bf16x3p3": (lambda t: split_bf16(t, 3), [(0, 0), (0, 1), (1, 0)]),
    "bf16x2p3": (lambda t: split_bf16(t, 2), [(0, 0), (0, 1), (1, 0)]),
}


def winograd_split_conv(x, w, b, split, terms):
    """3x3 conv, padding 1, as Winograd F(2x2,3x3) whose element-wise products run on split operands: V = B^T d B and
    U = G g G^T in f32 (as a kernel would form them), both split, the channel sums of the kept partial products in
    float64, Y = A^T M A in f32."""
    Bn, C, H, W = x.shape
    Hp, Wp = H + (H & 1), W + (W & 1)
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    pt = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # [B, C, th, tw, 4, 4]
    BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
    G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
    AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
    V = torch.einsum("ij,bctujk,lk->bctuil", BT, pt, BT)          # f32 adds only: exact order does not matter much here
    U = torch.einsum("ij,ocjk,lk->ocil", G, w.double(), G).float()
    Vs, Us = split(V), split(U)
    M = None
    for i, j in terms:
        t = torch.einsum("bctuil,ocil->botuil", Vs[i], Us[j])
        M = t if M is None else M + t
    M = M.float()
    Y = torch.einsum("ij,botujk,lk->botuil", AT, M, AT)           # [B, O, th, tw, 2, 2]
    th, tw = Y.shape[2], Y.shape[3]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, w.shape[0], 2 * th, 2 * tw)[:, :, :H, :W]
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y.contiguous()


def make_conv(scheme, only=None):
    wino = scheme.startswith("wino-")
    split, terms = SCHEMES[scheme[5:] if wino else scheme]

    def conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if groups != 1 or (only == "3x3" and w.shape[-1] != 3) or (only == "1x1" and w.shape[-1] != 1):
            return REAL(x, w, b, stride, padding, dilation, groups)
        STATS["max_x"] = max(STATS["max_x"], float(x.abs().max()))
        STATS["max_w"] = max(STATS["max_w"], float(w.abs().max()))
        if wino and w.shape[-1] == 3 and padding == 1 and w.shape[1] >= 16:
            return winograd_split_conv(x, w, b, split, terms)
        xs, ws = split(x), split(w)
        acc = None
        for i, j in terms:
            t = REAL(xs[i], ws[j], None, stride, padding, dilation, 1)
            acc = t if acc is None else acc + t
        if b is not None:
            acc = acc + b.double().view(1, -1, 1, 1)
        return acc.float()
    return conv


def run(name, scheme):
    stem, _, _ = LONG[name]
    g, seq = load_long(name)
    fut = int(g["args"][5])
    O.F.conv2d = REAL if scheme == "f32" else make_conv(scheme)
    try:
        orc = O.RecurrentOracle(load_weights(stem), future=fut)
        outs = orc.run_sequence(seq.raw, seq.flow_prev, seq.flow_next)
    finally:
        O.F.conv2d = REAL
    worst = max(float((outs[int(k)] - g["denoised"][i]).abs().max()) for i, k in enumerate(g["keep"]))
    dps = max(abs(O.psnr(outs[i][None], seq.gt[i + 1][None]) - float(g["PSNR"][i])) for i in range(outs.shape[0]))
    feat = float((orc.lastfeat[0] - g["feat_last"]).abs().max()) if "feat_last" in g else float("nan")
    return worst, dps, feat


def main():
    names = sys.argv[1:] or sorted(LONG)
    torch.set_num_threads(8)
    print(f"{'fixture':34s} {'scheme':9s} {'max|out-ref|':>13s} {'max dPSNR dB':>13s} {'max|feat-ref|':>14s}")
    for name in names:
        for scheme in ["f32"] + sorted(SCHEMES) + ["wino-f16x2p3", "wino-f16x2p4"]:
            worst, dps, feat = run(name, scheme)
            print(f"{name:34s} {scheme:9s} {worst:13.3e} {dps:13.2e} {feat:14.3e}", flush=True)
    print(f"largest |activation| into a dense conv {STATS['max_x']:.3f}, largest |weight| {STATS['max_w']:.3f}")


if __name__ == "__main__":
    main()
