#!/usr/bin/env python3
"""Writes a copy of rvdd-release_amd/csrc/convnext.hip whose convblock_pipe_kernel has timing-only switches (results WRONG on
purpose): -DRVDD_XP=<bits>, 1 no depth-wise taps (and none of their LDS reads), 2 no GELU (the hidden value is split as it is),
4 no MLP MFMAs (and none of their fragment reads), 8 no LayerNorm arithmetic (the sums written as they are).  What is left of a
tile's time, of the package power and of the clock with a part gone says what that part costs: tools/convblock_parts.sh.
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
rep('''split4h(gelu_phi4_scaled(hq[q][k], wt.gelu_c), hh[k], hl[k]);''',
    '''split4h((RVDD_XP & 2) ? hq[q][k] : gelu_phi4_scaled(hq[q][k], wt.gelu_c), hh[k], hl[k]);''')
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
out = sys.argv[1]
open(out, "w").write(head + body + tail)
print(out, "MFMA sites", n)
