#!/usr/bin/env python3
"""Writes a copy of rvdd-release_amd/csrc/convnext.hip whose convblock_pipe_kernel has timing-only switches (results WRONG on
purpose): -DRVDD_XP=<bits>, 1 no depth-wise taps (and none of their LDS reads), 2 no GELU (the hidden value is split as it is),
4 no MLP MFMAs (and none of their fragment reads), 8 no LayerNorm arithmetic (the sums written as they are).  What is left of a
tile's time, of the package power and of the clock with a part gone says what that part costs: tools/convblock_parts.sh.
Round 6, UPPER BOUNDS of the three structural variants of VERDICT r05 item 1 (what each could save at most, before any of it is built):
 16 the depth-wise tap weights from scalar registers instead of LDS (the 588 sixteen-byte weight reads per tile gone, the packed FMAs kept:
    what a channel-owning front whose weights are wave-uniform would save in LDS traffic, without what its lane map would cost);
 32 15 % fewer halo DMA pieces per chunk (28 of 33: what a 22x38 halo over a 16x32 tile would save in DMA bytes and LDS writes);
 64 the third halo chunk neither requested nor awaited (its taps run on whatever buffer 0 holds: MORE than a third buffer could hide --
    the chunk's traffic is gone too).
    python tools/convblock_xp_patch.py tools/scratch/convnext_xp.hip
The shipping kernel carries none of these."""
import os, sys

src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rvdd-release_amd", "csrc", "convnext.hip")
s = open(src).read()
a = s.index("void convblock_pipe_kernel(")
b = s.index("__global__ void pad_copy_kernel(")
head, body, tail = s[:a], s[a:b], s[b:]


def rep(x, y, n=1):
    global body
    assert body.count(x) == n, (body.count(x), x)
    body = body.replace(x, y)


rep('''                        for (int i = 0; i < 4; ++i) acc[i][j] = acc[i][j] + win[ky & 1][i + kx] * wv[ky & 1][kx];''',
    '''                        for (int i = 0; i < 4; ++i)
                            if (!(RVDD_XP & 1) || (ky == 3 && kx == 3)) acc[i][j] = acc[i][j] + win[ky & 1][i + kx] * wv[ky & 1][kx];''')
rep('''#define GS_(q, k, S) gelu_stage<S>(gs[k], hq[q][k], wt.gelu_c, hh[k], hl[k])''',
    '''#define GS_(q, k, S) do { if (!(RVDD_XP & 2)) gelu_stage<S>(gs[k], hq[q][k], wt.gelu_c, hh[k], hl[k]); else if (S == 5) split4h(hq[q][k], hh[k], hl[k]); } while (0)''')
n = body.count("__builtin_amdgcn_mfma_f32_16x16x32_f16(")
body = body.replace("__builtin_amdgcn_mfma_f32_16x16x32_f16(", "XP_MFMA(")
head += '''
#ifndef RVDD_XP
#define RVDD_XP 0
#endif
#define XP_MFMA(a, b, c, x, y, z) ((RVDD_XP & 4) ? (c) : __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z))
'''
rep('''                    for (int k = 0; k < 4; ++k) r[k] = lw[j][k] * ((acc[i][j][k] - u) * rden) + lb[j][k];
                    *reinterpret_cast<f32x4*>(Xw + x_wr + j * 256 + i * 16) = r;''',
    '''                    for (int k = 0; k < 4; ++k) r[k] = (RVDD_XP & 8) ? acc[i][j][k] : lw[j][k] * ((acc[i][j][k] - u) * rden) + lb[j][k];
                    *reinterpret_cast<f32x4*>(Xw + x_wr + j * 256 + i * 16) = r;''')
rep('''                    for (int kx = 0; kx < 7; ++kx) ww[kx] = *reinterpret_cast<const f32x4*>(wb + (ky * 7 + kx) * kF);''',
    '''                    for (int kx = 0; kx < 7; ++kx) {
                        if (RVDD_XP & 16) ww[kx] = f32x4{wt.gelu_c[0][0], wt.gelu_c[1][0], wt.gelu_c[2][0], wt.gelu_c[3][0]};      // (four scalars in all: 28 distinct ones per filter row spilled SGPRs into lanes, 800 v_readlane / v_writelane)
                        else ww[kx] = *reinterpret_cast<const f32x4*>(wb + (ky * 7 + kx) * kF);
                    }''')
rep('''                if (k < E_PIECES) {
                    const int gy = tp.y0 - 3 + (piece_yx[n] >> 8), gx = tp.x0 - 3 + (piece_yx[n] & 255);''',
    '''                if (k < ((RVDD_XP & 32) ? 28 : E_PIECES)) {
                    const int gy = tp.y0 - 3 + (piece_yx[n] >> 8), gx = tp.x0 - 3 + (piece_yx[n] & 255);''')
rep('''                if (j == 0) {
                    fsync();                         // every front wave is done with buffer 0
                    dma_chunk(cur, 2, 0);
                }''',
    '''                if (j == 0 && !(RVDD_XP & 64)) {
                    fsync();                         // every front wave is done with buffer 0
                    dma_chunk(cur, 2, 0);
                }''')
rep('''                if (j != 1) fsync();
                PSTAMP(1);''',
    '''                if (j == 0 || (j == 2 && !(RVDD_XP & 64))) fsync();
                PSTAMP(1);''')
out = sys.argv[1]
open(out, "w").write(head + body + tail)
print(out, "MFMA sites", n)
