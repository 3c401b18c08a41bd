"""The validation loop (validate.py:54-114) on the HIP runtime: two videos back to back, with the
dataset's flows and with flows recomputed online from the previous denoised frame
(validate.py:16-38, TV-L1 on the device), against the two CPU oracles chained the same way."""
import os

import numpy as np
import pytest
import torch

import rvdd_oracle as O
import tvl1_oracle as T
from conftest import WEIGHTS, load_weights

pytestmark = pytest.mark.gpu

STEM = "recurrent-convunet+feat-iso3200"
NET = "convunet-mode=fixedfeatures+feat"


def _dataset(seqs):
    for v, s in enumerate(seqs):
        for t in range(1, s.raw.shape[0]):
            yield {"n": torch.cat((s.raw[t - 1], s.raw[t]), 0)[None], "flow": s.flow_prev[t][None, None],
                   "gt": torch.cat((s.gt[t - 1], s.gt[t]), 0)[None],
                   "n_path": [f"video{v}/{t:03d}.tif"], "gt_path": [f"video{v}/{t:03d}.tif"]}


def _remosaick(x):
    y = torch.zeros(x.shape[0], 4, x.shape[2] // 2, x.shape[3] // 2)
    y[:, 0], y[:, 1] = x[:, 1, 0::2, 0::2], x[:, 2, 0::2, 1::2]
    y[:, 2], y[:, 3] = x[:, 0, 1::2, 0::2], x[:, 1, 1::2, 1::2]
    return y


def _oracle_loop(sd, seqs, online, lam, flow_from=None):
    """flow_from: the frames the online flows are computed from instead of the oracle's own outputs (one per output
    frame, in order).  TV-L1 stops each warp when its update falls under a threshold (tvl1flow_lib.c:205): two
    previous frames 1e-6 apart can stop one iteration apart and give flows 0.2 px apart (tools/online_flow_sensitivity.py
    shows exactly that between two of this library's own conv kernels), so a frame-by-frame comparison of two
    free-running loops measures that coin toss, not the denoiser.  Handing the oracle the frames the device loop
    computed its flows from removes the toss and keeps everything else: TV-L1 on both sides, warp, net, recurrence."""
    outs, l1s, psnrs = [], [], []
    for s in seqs:
        rec = O.RecurrentOracle(sd, future=0)
        den = None
        for t in range(1, s.raw.shape[0]):
            flow = s.flow_prev[t][None]
            if online and t > 1:
                a = ((s.raw[t] + 1) / 2).permute(1, 2, 0).numpy()
                prev = den if flow_from is None else flow_from[len(outs) - 1][None]
                b = ((_remosaick(prev)[0] + 1) / 2).permute(1, 2, 0).numpy()
                flow = torch.from_numpy(T.TVL1_flow(a, b).transpose(2, 0, 1).copy())[None]
            den = rec.step(s.raw[t - 1][None], s.raw[t][None], None, flow, None, first=(t == 1))
            outs.append(den[0])
            l1s.append(O.l1_loss(den, s.gt[t][None], lam))
            psnrs.append(O.psnr(den, s.gt[t][None]))
    return outs, float(np.mean(l1s)), float(np.mean(psnrs))


@pytest.mark.parametrize("online", [False, True])
def test_compute_validation(online):
    from rvdd_release_amd import synth
    from rvdd_release_amd.models import create_model
    from rvdd_release_amd.options import make_opt
    from rvdd_release_amd.validate import compute_validation
    sd = load_weights(STEM)
    seqs = [synth.make_sequence(4, 96, 128, iso=3200, seed=60 + v) for v in range(2)]
    opt = make_opt(netDenoiser=NET, feature_rec=True, future_patch_depth=0,
                   path2epoch=os.path.join(WEIGHTS, STEM), gpu_ids=[0], val_flow_from_denoised=online)
    model = create_model(opt)
    model.setup(opt)
    opt.isTrain = False
    model.isTrain = False
    got = []
    res = compute_validation(model, _dataset(seqs), opt,
                             on_frame=lambda i, d, vis, l: got.append((vis["denoised"][0].cpu(), d["FirstOfVideo"])))
    # online: the oracle computes each flow from the frame the device loop computed it from (see _oracle_loop)
    want, l1, psnr = _oracle_loop(sd, seqs, online, opt.lambda_L1, [g for g, _ in got] if online else None)
    assert [f for _, f in got] == [True, False, False, True, False, False]
    # online flows differ from the oracle's by up to ~2e-3 px (tests/test_tvl1.py); through the bicubic
    # warp of a [-1,1] image that is well below 2e-3 in the output
    tol = 2e-3 if online else 1e-4
    for (g, _), w in zip(got, want):
        assert (g - w).abs().max() < tol, float((g - w).abs().max())
    assert set(res) == {"L1_valLoss", "PSNR_valLoss", "Denoiser_valLoss", "lr"}
    assert abs(res["PSNR_valLoss"] - psnr) < 0.02
    assert abs(res["L1_valLoss"] - l1) < 1e-3
    assert res["lr"] == opt.lr and model.isTrain is False


def test_compute_validation_online_flow_with_future_frame():
    """--val_flow_from_denoised with --future_patch_depth 1: the flow towards the previous frame is recomputed
    from the previous OUTPUT (TV-L1 on the device), the flow towards the next frame stays the dataset's."""
    from rvdd_release_amd import synth
    from rvdd_release_amd.models import create_model
    from rvdd_release_amd.options import make_opt
    from rvdd_release_amd.validate import compute_validation
    stem = "recurrent-convunet+feat-future-iso12800"
    sd = load_weights(stem)
    s = synth.make_sequence(5, 96, 128, iso=12800, seed=70)

    def dataset():
        for t in range(1, s.raw.shape[0] - 1):
            yield {"n": torch.cat((s.raw[t - 1], s.raw[t], s.raw[t + 1]), 0)[None],
                   "flow": torch.stack((s.flow_prev[t], s.flow_next[t]), 0)[None],
                   "gt": torch.cat((s.gt[t - 1], s.gt[t]), 0)[None],
                   "n_path": [f"video0/{t:03d}.tif"], "gt_path": [f"video0/{t:03d}.tif"]}

    opt = make_opt(netDenoiser=NET, feature_rec=True, future_patch_depth=1, path2epoch=os.path.join(WEIGHTS, stem),
                   gpu_ids=[0], val_flow_from_denoised=True)
    model = create_model(opt)
    model.setup(opt)
    opt.isTrain = model.isTrain = False
    got = []
    compute_validation(model, dataset(), opt, on_frame=lambda i, d, vis, l: got.append(vis["denoised"][0].cpu()))
    rec = O.RecurrentOracle(sd, future=1)
    den = None
    for k, t in enumerate(range(1, s.raw.shape[0] - 1)):
        flow = s.flow_prev[t][None]
        if t > 1:
            a = ((s.raw[t] + 1) / 2).permute(1, 2, 0).numpy()                    # the CURRENT noisy frame
            b = ((_remosaick(den)[0] + 1) / 2).permute(1, 2, 0).numpy()
            flow = torch.from_numpy(T.TVL1_flow(a, b).transpose(2, 0, 1).copy())[None]
        den = rec.step(s.raw[t - 1][None], s.raw[t][None], s.raw[t + 1][None], flow, s.flow_next[t][None], first=(t == 1))
        assert (got[k] - den[0]).abs().max() < 2e-3, (t, float((got[k] - den[0]).abs().max()))
    assert len(got) == 3
