"""CPU oracle for the TV-L1 optical flow that the reference pre-computes its flows with
(`libBridge.cpp:44-163` -> `3rdparty/tvl1flow/*.c`, IPOL Dual TV-L1).

TEST INFRASTRUCTURE ONLY (see oracle/rvdd_oracle.py for the rules).  NumPy restatement, every
function citing the C lines it follows, including the quirks a reader would "fix":
  * `bicubic_interpolation_at` derives the previous ROW from the sign of the COLUMN coordinate
    (`my = bc((int) vv - sx)`, bicubic_interpolation.c:165 -- sx, not sy);
  * truncation `(int) uu` is toward zero, so coordinates in (-1, 0) use pixel 0 as base;
  * the Gaussian's reflecting boundary repeats the edge sample on the right/bottom but not on the
    left/top (mask.c:275-279, 311-315), and its constant is 3.1415926 (mask.c:257);
  * `hypot` runs in double and is rounded to float (tvl1flow_lib.c:224-225).

Parity status: PINNED against the reference itself compiled from its own sources
(`oracle/Makefile` -> `oracle/_ref/libBridge.so`): tests/test_tvl1_oracle.py checks this file
against it when the library is present, and against golden flows generated with it
(tests/golden/tvl1_*.npz, tools/make_golden_tvl1.py) everywhere else.  The reference sums its
convergence error with an OpenMP reduction, so its iteration count near the threshold -- and hence
its flow in the last digits -- depends on the thread count; comparisons are by tolerance.
"""
from __future__ import annotations

import math

import numpy as np

F = np.float32
TAU, LAMBDA, THETA, ZFACTOR, EPSILON = F(0.25), F(0.15), F(0.3), F(0.5), F(0.01)   # libBridge.cpp:27-36
NWARPS, MAX_ITERATIONS, NSCALES_MAX = 5, 300, 100
PRESMOOTHING_SIGMA, GRAD_IS_ZERO, ZOOM_SIGMA_ZERO = 0.8, 1e-10, 0.6


# ---- bicubic_interpolation.c --------------------------------------------------------------
def _cubic(v0, v1, v2, v3, x):
    """cubic_interpolation_cell (bicubic_interpolation.c:95-103), double."""
    return v1 + 0.5 * x * (v2 - v0 + x * (2.0 * v0 - 5.0 * v1 + 4.0 * v2 - v3 + x * (3.0 * (v1 - v2) + v3 - v0)))


def bicubic_at(img: np.ndarray, uu: np.ndarray, vv: np.ndarray, border_out: bool) -> np.ndarray:
    """bicubic_interpolation_at (bicubic_interpolation.c:133-231), BOUNDARY_CONDITION 0 (neumann)."""
    ny, nx = img.shape
    uu = uu.astype(F)
    vv = vv.astype(F)
    sx = np.where(uu < 0, -1, 1)
    sy = np.where(vv < 0, -1, 1)
    iu = np.trunc(uu).astype(np.int64)
    iv = np.trunc(vv).astype(np.int64)
    out = np.zeros(uu.shape, bool)

    def bc(v, n):
        nonlocal out
        out |= (v < 0) | (v >= n)
        return np.clip(v, 0, n - 1)

    x, y = bc(iu, nx), bc(iv, ny)
    mx, my = bc(iu - sx, nx), bc(iv - sx, ny)          # :164-165  (my uses sx)
    dx, dy = bc(iu + sx, nx), bc(iv + sy, ny)
    ddx, ddy = bc(iu + 2 * sx, nx), bc(iv + 2 * sy, ny)
    p = lambda r, c: img[r, c].astype(np.float64)
    fx = (uu - x.astype(F)).astype(np.float64)           # uu - x in float, widened by the call
    fy = (vv - y.astype(F)).astype(np.float64)
    cols = []
    for c in (mx, x, dx, ddx):                            # pol[i] = column i, interpolated along y first
        cols.append(_cubic(p(my, c), p(y, c), p(dy, c), p(ddy, c), fy))
    res = _cubic(cols[0], cols[1], cols[2], cols[3], fx).astype(F)
    if border_out:
        res = np.where(out, F(0), res)
    return res


def warp(img, u, v, border_out=True):
    """bicubic_interpolation_warp (bicubic_interpolation.c:240-262)."""
    ny, nx = img.shape
    jj, ii = np.meshgrid(np.arange(nx, dtype=F), np.arange(ny, dtype=F))
    return bicubic_at(img, (jj + u).astype(F), (ii + v).astype(F), border_out)


# ---- mask.c ---------------------------------------------------------------------------------
def gaussian(I: np.ndarray, sigma: float) -> np.ndarray:
    """In-place Gaussian of mask.c:236-329 (reflecting boundary, double accumulation)."""
    ydim, xdim = I.shape
    I = I.astype(F).copy()
    den = 2 * sigma * sigma
    size = int(5 * sigma) + 1
    if size > xdim:
        raise ValueError("GaussianSmooth: sigma too large")
    B = np.array([1 / (sigma * math.sqrt(2.0 * 3.1415926)) * math.exp(-i * i / den) for i in range(size)])
    norm = B.sum() * 2 - B[0]
    B = B / norm

    def conv_lines(A):        # A [lines, n] float32 -> filtered float32, each line independently
        n = A.shape[1]
        R = np.zeros((A.shape[0], n + 2 * size))
        R[:, size:size + n] = A
        for i in range(size):
            R[:, i] = A[:, size - i]                      # left:  I[size-i]      (no edge repeat)
            R[:, size + n + i] = A[:, n - i - 1]          # right: I[n-i-1]       (edge repeated)
        s = B[0] * R[:, size:size + n]
        for j in range(1, size):
            s = s + B[j] * (R[:, size - j:size - j + n] + R[:, size + j:size + j + n])
        return s.astype(F)

    I = conv_lines(I)
    I = conv_lines(I.T.copy()).T.copy()
    return I


def centered_gradient(I):
    """mask.c:148-206."""
    I = I.astype(F)
    dx = np.empty_like(I)
    dy = np.empty_like(I)
    h = F(0.5)
    dx[:, 1:-1] = h * (I[:, 2:] - I[:, :-2])
    dx[:, 0] = h * (I[:, 1] - I[:, 0])
    dx[:, -1] = h * (I[:, -1] - I[:, -2])
    dy[1:-1, :] = h * (I[2:, :] - I[:-2, :])
    dy[0, :] = h * (I[1, :] - I[0, :])
    dy[-1, :] = h * (I[-1, :] - I[-2, :])
    return dx, dy


def forward_gradient(f):
    """mask.c:98-140."""
    fx = np.zeros_like(f)
    fy = np.zeros_like(f)
    fx[:, :-1] = f[:, 1:] - f[:, :-1]
    fy[:-1, :] = f[1:, :] - f[:-1, :]
    return fx, fy


def divergence(v1, v2):
    """mask.c:40-90 (backward differences with the boundary rows/columns of the paper)."""
    d = np.zeros_like(v1)
    v1x = np.zeros_like(v1)
    v2y = np.zeros_like(v1)
    v1x[:, 1:-1] = v1[:, 1:-1] - v1[:, :-2]
    v1x[:, 0] = v1[:, 0]
    v1x[:, -1] = -v1[:, -2]
    v2y[1:-1, :] = v2[1:-1, :] - v2[:-2, :]
    v2y[0, :] = v2[0, :]
    v2y[-1, :] = -v2[-2, :]
    d = v1x + v2y
    return d.astype(F)


# ---- zoom.c ---------------------------------------------------------------------------------
def zoom_size(nx, ny, factor):
    return int(F(nx) * F(factor) + F(0.5)), int(F(ny) * F(factor) + F(0.5))


def zoom_out(I, factor):
    """zoom.c:41-78."""
    ny, nx = I.shape
    nxx, nyy = zoom_size(nx, ny, factor)
    sigma = float(F(ZOOM_SIGMA_ZERO * math.sqrt(1.0 / (float(factor) * float(factor)) - 1.0)))
    Is = gaussian(I, sigma)
    jj, ii = np.meshgrid(np.arange(nxx, dtype=F), np.arange(nyy, dtype=F))
    return bicubic_at(Is, (jj / F(factor)).astype(F), (ii / F(factor)).astype(F), False)


def zoom_in(I, nxx, nyy):
    """zoom.c:85-108."""
    ny, nx = I.shape
    fx, fy = F(nxx) / F(nx), F(nyy) / F(ny)
    jj, ii = np.meshgrid(np.arange(nxx, dtype=F), np.arange(nyy, dtype=F))
    return bicubic_at(I, (jj / fx).astype(F), (ii / fy).astype(F), False)


# ---- tvl1flow_lib.c -------------------------------------------------------------------------
def dual_tvl1_one_scale(I0, I1, u1, u2, stats=None):
    """Dual_TVL1_optic_flow (tvl1flow_lib.c:91-278) with the libBridge parameters."""
    l_t = F(LAMBDA * THETA)
    taut = F(TAU / THETA)
    eps2 = F(EPSILON * EPSILON)
    I1x, I1y = centered_gradient(I1)
    p11 = np.zeros_like(I0); p12 = np.zeros_like(I0); p21 = np.zeros_like(I0); p22 = np.zeros_like(I0)
    size = I0.size
    for _ in range(NWARPS):
        I1w = warp(I1, u1, u2, True)
        I1wx = warp(I1x, u1, u2, True)
        I1wy = warp(I1y, u1, u2, True)
        grad = (I1wx * I1wx + I1wy * I1wy).astype(F)
        rho_c = (I1w - I1wx * u1 - I1wy * u2 - I0).astype(F)
        n = 0
        error = np.inf
        while error > eps2 and n < MAX_ITERATIONS:
            n += 1
            rho = (rho_c + (I1wx * u1 + I1wy * u2)).astype(F)
            lg = (l_t * grad).astype(F)
            with np.errstate(divide="ignore", invalid="ignore"):
                fi = (-rho / grad).astype(F)
                fi = np.where(grad < F(GRAD_IS_ZERO), F(0), fi)     # unused there (the branch yields 0)
            d1 = np.where(rho < -lg, l_t * I1wx, np.where(rho > lg, -l_t * I1wx,
                          np.where(grad < F(GRAD_IS_ZERO), F(0), fi * I1wx))).astype(F)
            d2 = np.where(rho < -lg, l_t * I1wy, np.where(rho > lg, -l_t * I1wy,
                          np.where(grad < F(GRAD_IS_ZERO), F(0), fi * I1wy))).astype(F)
            v1 = (u1 + d1).astype(F)
            v2 = (u2 + d2).astype(F)
            div1 = divergence(p11, p12)
            div2 = divergence(p21, p22)
            u1n = (v1 + THETA * div1).astype(F)
            u2n = (v2 + THETA * div2).astype(F)
            e = ((u1n - u1) * (u1n - u1) + (u2n - u2) * (u2n - u2)).astype(F)
            error = F(e.sum(dtype=np.float64) / size)
            u1, u2 = u1n, u2n
            u1x, u1y = forward_gradient(u1)
            u2x, u2y = forward_gradient(u2)
            g1 = np.hypot(u1x.astype(np.float64), u1y.astype(np.float64)).astype(F)
            g2 = np.hypot(u2x.astype(np.float64), u2y.astype(np.float64)).astype(F)
            ng1 = (1.0 + (taut * g1).astype(np.float64)).astype(F)
            ng2 = (1.0 + (taut * g2).astype(np.float64)).astype(F)
            p11 = ((p11 + taut * u1x) / ng1).astype(F)
            p12 = ((p12 + taut * u1y) / ng1).astype(F)
            p21 = ((p21 + taut * u2x) / ng2).astype(F)
            p22 = ((p22 + taut * u2y) / ng2).astype(F)
        if stats is not None:
            stats.append(n)
    return u1, u2


def num_scales(nx, ny):
    """libBridge.cpp:131-136."""
    N = 1 + math.log(math.hypot(nx, ny) / 16.0) / math.log(float(F(1) / ZFACTOR))
    return max(1, min(NSCALES_MAX, int(N)))


def tvl1flow(I0: np.ndarray, I1: np.ndarray, stats=None) -> np.ndarray:
    """`tvl1flow(I0, I1, u, nx, ny)` (libBridge.cpp:44-163): [ny,nx] x2 -> flow [2,ny,nx] (u then v).
    Dual_TVL1_optic_flow_multiscale, tvl1flow_lib.c:343-472."""
    ny, nx = I0.shape
    I0 = I0.astype(F)
    I1 = I1.astype(F)
    ns = num_scales(nx, ny)
    mn = min(I0.min(), I1.min())
    mx = max(I0.max(), I1.max())
    den = F(mx - mn)
    if den > 0:                                                           # image_normalization :300-333
        I0 = (255.0 * (I0.astype(np.float64) - float(mn)) / float(den)).astype(F)
        I1 = (255.0 * (I1.astype(np.float64) - float(mn)) / float(den)).astype(F)
    I0s = [gaussian(I0, PRESMOOTHING_SIGMA)]
    I1s = [gaussian(I1, PRESMOOTHING_SIGMA)]
    for s in range(1, ns):
        I0s.append(zoom_out(I0s[-1], ZFACTOR))
        I1s.append(zoom_out(I1s[-1], ZFACTOR))
    u1 = np.zeros_like(I0s[-1])
    u2 = np.zeros_like(I0s[-1])
    for s in range(ns - 1, -1, -1):
        u1, u2 = dual_tvl1_one_scale(I0s[s], I1s[s], u1, u2, stats)
        if s == 0:
            break
        nyy, nxx = I0s[s - 1].shape
        u1 = (zoom_in(u1, nxx, nyy) * (F(1.0) / ZFACTOR)).astype(F)
        u2 = (zoom_in(u2, nxx, nyy) * (F(1.0) / ZFACTOR)).astype(F)
    return np.stack((u1, u2), 0)


def TVL1_flow(Im1: np.ndarray, Im2: np.ndarray) -> np.ndarray:
    """library.CPPbridge.TVL1_flow (library.py:150-175) for the 4-channel raw case: channel mean,
    returns [h,w,2]."""
    assert Im1.shape == Im2.shape and Im1.shape[2] in (1, 4)
    g1 = Im1.mean(axis=2).astype(F) if Im1.shape[2] == 4 else Im1[..., 0].astype(F)
    g2 = Im2.mean(axis=2).astype(F) if Im2.shape[2] == 4 else Im2[..., 0].astype(F)
    return tvl1flow(g1, g2).transpose(1, 2, 0)
