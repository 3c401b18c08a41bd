"""The oracle (oracle/rvdd_oracle.py) against the fixtures that
tools/make_golden.py captured from the reference itself.  CPU only.

Tolerances: the Hamilton-Adams restatement must be bit-exact (hard sign()
selections); everything else is fp32 re-association noise, bounded here by
max-abs 2e-5 on O(1) data (observed <= 4e-6)."""
import os

import numpy as np
import pytest
import torch

import rvdd_oracle as O
from conftest import GOLDEN, LONG, VARIANTS, load_long, load_weights


def _npz(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, name)).items()}


def test_hamilton_adams_bit_exact():
    g = _npz("op_hamilton_adams.npz")
    rgb = O.hamilton_adams(g["raw"])
    assert rgb.shape == g["rgb"].shape
    assert torch.equal(rgb, g["rgb"])
    assert torch.equal(O.remosaick(rgb[:, :3]), g["remosaick"])


def test_warp_bicubic():
    g = _npz("op_warp_bicubic.npz")
    assert (O.warp(g["x"], g["flow"]) - g["y"]).abs().max() < 1e-6
    # the explicit 16-tap statement the HIP kernel follows
    assert (O.warp_explicit(g["x"], g["flow"]) - g["y"]).abs().max() < 2e-5


def test_upsample_flow():
    g = _npz("op_upsample_flow.npz")
    assert torch.equal(O.upsample_factor_2(g["flow"], 2), g["up"])
    assert (O.upsample_factor_2_explicit(g["flow"], 2) - g["up"]).abs().max() < 1e-5


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_net_forward(name):
    stem, fut, _ = VARIANTS[name]
    sd = load_weights(stem)
    g = _npz(f"net_{name}.npz")
    for tag in ("20x28", "16x24"):
        fin = g.get(f"feat_in_{tag}")
        out, f = O.net_forward(sd, g[f"x_{tag}"], fin)
        assert (out - g[f"out_{tag}"]).abs().max() < 2e-5
        if fin is not None:
            assert (f - g[f"feat_out_{tag}"]).abs().max() < 2e-5


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_sequence(name):
    stem, fut, _ = VARIANTS[name]
    sd = load_weights(stem)
    g = _npz(f"seq_{name}.npz")
    orc = O.RecurrentOracle(sd, future=fut)
    outs = orc.run_sequence(g["raw"], g["flow_prev"], g["flow_next"])
    assert outs.shape == g["denoised"].shape
    assert (outs - g["denoised"]).abs().max() < 2e-5
    if "feat_last" in g:
        assert (orc.lastfeat[0] - g["feat_last"]).abs().max() < 2e-5
    for i in range(outs.shape[0]):
        gt = g["gt"][i + 1][None]
        assert abs(O.psnr(outs[i][None], gt) - float(g["PSNR"][i])) < 1e-3
        assert abs(O.l1_loss(outs[i][None], gt) - float(g["L1"][i])) < 1e-4


@pytest.mark.parametrize("name,stem,fut", [("nowarp-iso3200", "non_recurrent-convunet-no_warp-iso3200", 0),
                                           ("nowarp-future-iso3200", "non_recurrent-convunet-no_warp-future-iso3200", 1)])
def test_no_warp_sequence_matches_reference(name, stem, fut):
    """--no_warp (scripts/test-non_recurrent-no_warp-*.sh): previous output and next frame enter the net unwarped."""
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, f"seq_{name}.npz")).items()}
    rec = O.RecurrentOracle(load_weights(stem), future=fut, no_warp=True)
    T = g["raw"].shape[0]
    for k, t in enumerate(range(1, T - fut)):
        den = rec.step(g["raw"][t - 1][None], g["raw"][t][None], g["raw"][t + 1][None] if fut else None, None, None, first=(t == 1))
        assert (den[0] - g["denoised"][k]).abs().max() < 2e-5
        assert abs(O.psnr(den, g["gt"][t][None]) - float(g["PSNR"][k])) < 1e-3


def test_prev_noisy_frame_sequence_matches_reference():
    """--prev_noisy_frame: the previous frame of step t+1 is the demosaiced noisy frame t (features stay recurrent)."""
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "seq_prevnoisy-feat-iso3200.npz")).items()}
    rec = O.RecurrentOracle(load_weights("recurrent-convunet+feat-iso3200"), future=0, prev_noisy_frame=True)
    for k, t in enumerate(range(1, g["raw"].shape[0])):
        den = rec.step(g["raw"][t - 1][None], g["raw"][t][None], None, g["flow_prev"][t][None], None, first=(t == 1))
        assert (den[0] - g["denoised"][k]).abs().max() < 2e-5


@pytest.mark.parametrize("name,stem,fut", [("warpraw-iso3200", "recurrent-convunet-iso3200", 0),
                                           ("warpraw-future-iso3200", "recurrent-convunet-future-iso3200", 1)])
def test_warp_raw_sequence_matches_reference(name, stem, fut):
    """--warp_raw: frames are re-mosaicked, warped at raw resolution with the raw-resolution flow, demosaicked again."""
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, f"seq_{name}.npz")).items()}
    rec = O.RecurrentOracle(load_weights(stem), future=fut, warp_raw=True)
    T = g["raw"].shape[0]
    for k, t in enumerate(range(1, T - fut)):
        den = rec.step(g["raw"][t - 1][None], g["raw"][t][None], g["raw"][t + 1][None] if fut else None,
                       g["flow_prev"][t][None], g["flow_next"][t][None] if fut else None, first=(t == 1))
        assert (den[0] - g["denoised"][k]).abs().max() < 2e-5


@pytest.mark.parametrize("name", sorted(LONG))
def test_whole_sequence_matches_reference(name):
    """30 / 90 dependent steps (models/recurrent_model.py:335-345 feeds each output back): the oracle against the
    reference's own run over BASELINE's sequence lengths (tools/make_golden_long.py) -- stored frames, last features,
    and the reference's L1 / PSNR of EVERY frame."""
    stem, _, _ = LONG[name]
    g, seq = load_long(name)
    fut = int(g["args"][5])
    orc = O.RecurrentOracle(load_weights(stem), future=fut)
    outs = orc.run_sequence(seq.raw, seq.flow_prev, seq.flow_next)
    assert outs.shape[0] == g["PSNR"].shape[0]
    curve = [float((outs[int(k)] - g["denoised"][i]).abs().max()) for i, k in enumerate(g["keep"])]
    assert max(curve) < 2e-5, curve
    assert (orc.lastfeat[0] - g["feat_last"]).abs().max() < 5e-5
    for i in range(outs.shape[0]):
        gt = seq.gt[i + 1][None]
        assert abs(O.psnr(outs[i][None], gt) - float(g["PSNR"][i])) < 1e-3, i
        assert abs(O.l1_loss(outs[i][None], gt) - float(g["L1"][i])) < 1e-4, i
