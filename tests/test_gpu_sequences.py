"""Whole-sequence parity of the HIP path: the recurrence over the sequence lengths BASELINE.json quotes
(30 frames for C2 / C3 / C4, 90 for C5; models/recurrent_model.py:335-345 feeds every output back, so an error
made at frame t is an input of every later frame), the per-GPU share of config C5 at its real size, and the
collectives of the multi-GPU job on RCCL at world size 1.  Needs a real MI355X: -m gpu.

Bars: max-abs < 1e-4 on EVERY frame (the per-frame curve is in the assertion message), task PSNR within 0.01 dB
of the reference's / the oracle's on the same frame, L1 within 1e-3.
"""
import json
import math
import os
import socket
import subprocess
import sys

import pytest
import torch

import rvdd_oracle as O
from conftest import LONG, REPO, WEIGHTS, load_long, load_weights

pytestmark = pytest.mark.gpu

NETSTR = {"convunet+feat": "convunet-mode=fixedfeatures+feat", "next+feat": "newunet-mode=feat"}


def parity_psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 200.0 if mse == 0 else 10 * math.log10(4.0 / mse)


@pytest.mark.parametrize("name", sorted(LONG))
def test_whole_sequence_golden_model_surface(name):
    """The reference's own 30- / 90-frame runs (tools/make_golden_long.py), driven through the model surface exactly
    like validate.py:64-88: the reference's stored frames, its L1 and PSNR of EVERY frame, its last features."""
    from rvdd_release_amd.models import create_model
    from rvdd_release_amd.options import make_opt
    stem, arch, _ = LONG[name]
    g, seq = load_long(name)
    T, fut = int(g["args"][0]), int(g["args"][5])
    opt = make_opt(netDenoiser=NETSTR[arch], feature_rec=True, future_patch_depth=fut,
                   path2epoch=os.path.join(WEIGHTS, stem), gpu_ids=[0])
    model = create_model(opt)
    model.setup(opt)
    opt.isTrain = model.isTrain = False
    model.eval()
    keep = {int(k): i for i, k in enumerate(g["keep"])}
    curve, dpsnr = {}, []
    for t in range(1, T - fut):
        frames = [seq.raw[t - 1], seq.raw[t]] + ([seq.raw[t + 1]] if fut else [])
        flows = [seq.flow_prev[t]] + ([seq.flow_next[t]] if fut else [])
        data = {"n": torch.cat(frames, 0)[None], "flow": torch.stack(flows, 0)[None],
                "gt": torch.cat((seq.gt[t - 1], seq.gt[t]), 0)[None], "n_path": [f"seq/{t:03d}.tif"],
                "gt_path": [f"seq/{t:03d}.tif"], "FirstOfVideo": t == 1}
        model.set_input(data)
        model.test()
        model.compute_losses()
        losses = model.get_current_losses()
        dpsnr.append(abs(losses["PSNR"] - float(g["PSNR"][t - 1])))
        assert dpsnr[-1] < 0.01, (t, losses["PSNR"], float(g["PSNR"][t - 1]))
        assert abs(losses["L1"] - float(g["L1"][t - 1])) < 1e-3, t
        if t - 1 in keep:
            den = model.get_current_visuals()["denoised"][0].cpu()
            curve[t] = float((den - g["denoised"][keep[t - 1]]).abs().max())
    assert len(dpsnr) == T - 1 - fut and max(curve.values()) < 1e-4, f"max-abs vs the reference per stored frame: {curve}"
    _, feat = model._rt.get_state()
    assert (feat[0].cpu() - g["feat_last"]).abs().max() < 2e-4


@pytest.mark.parametrize("cfg,arch,stem,fut,iso,T", [
    ("C2", "convunet+feat", "recurrent-convunet+feat-iso3200", 0, 3200, 30),
    ("C3", "convunet+feat", "recurrent-convunet+feat-future-iso12800", 1, 12800, 30),
    ("C4", "next+feat", "recurrent-ConvNeXtUnet+feat-future-iso3200", 1, 3200, 30),
    ("C5", "convunet+feat", "recurrent-convunet+feat-iso3200", 0, 3200, 90),
])
def test_whole_sequence_vs_oracle(cfg, arch, stem, fut, iso, T):
    """Every frame of a whole sequence (BASELINE's lengths) against the oracle, B = 2 in lockstep at 180x320: a size
    where the 1/8 level (22x40) is ragged, the 45-row level is zero-padded against its skip and both conv kernels
    take part (Winograd at the two fine levels, direct below)."""
    from rvdd_release_amd import synth
    from rvdd_release_amd.runtime import RvddRuntime
    sd = load_weights(stem)
    B, H, W = 2, 180, 320
    seqs = [synth.make_sequence(T, H, W, iso=iso, seed=8000 + 10 * int(cfg[1]) + b) for b in range(B)]
    rt = RvddRuntime(arch, fut, B, H, W, 0)
    rt.load_state_dict(sd)
    raw = torch.stack([s.raw for s in seqs], 1).cuda()                # [T,B,4,h,w]
    fp = torch.stack([s.flow_prev for s in seqs], 1).cuda()
    fn = torch.stack([s.flow_next for s in seqs], 1).cuda()
    n_out = T - 1 - fut
    got = torch.empty(n_out, B, 3, H, W, device="cuda")
    for t in range(1, T - fut):
        rt.step(raw[t - 1] if t == 1 else None, raw[t], raw[t + 1] if fut else None, fp[t], fn[t] if fut else None, out=got[t - 1])
    got = got.cpu()
    _, feat = rt.get_state()
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    for b in range(B):
        orc = O.RecurrentOracle(sd, future=fut)
        want = orc.run_sequence(seqs[b].raw, seqs[b].flow_prev, seqs[b].flow_next)
        curve = [float((got[k, b] - want[k]).abs().max()) for k in range(n_out)]
        msg = f"{cfg} seq {b}: max-abs per frame " + " ".join(f"{v:.1e}" for v in curve)
        assert max(curve) < 1e-4, msg
        assert parity_psnr(got[-1, b], want[-1]) > 110.0, msg
        for k in (0, n_out // 2, n_out - 1):                          # task PSNR, the last frame included
            gt = seqs[b].gt[k + 1][None]
            assert abs(O.psnr(got[k, b][None], gt) - O.psnr(want[k][None], gt)) < 0.01, (k, msg)
        assert (feat[b].cpu() - orc.lastfeat[0]).abs().max() < 2e-4, msg
    rt.close()


def test_fused_upsample_at_bench_size_over_a_sequence():
    """UpConv's fused upsample against the two-kernel form at the BENCH shape (8 sequences of 1280x720, 12 frames): every frame and the
    recurrent features bit for bit.  The small shapes of test_fused_upsample_equals_upsample_then_conv give every workgroup of the conv one
    tile; round 6 had a build whose workgroups staged wrong pixels in the first trip of their tile loop when the tile fetched there was a
    border tile -- only a launch with several tiles per workgroup shows that (profiles/r06s_upsample_nondeterminism.md; the rare form of
    the same defect takes tools/determinism_soak.py)."""
    from rvdd_release_amd import synth
    from rvdd_release_amd.runtime import RvddRuntime
    sd = load_weights("recurrent-convunet+feat-iso3200")
    B, H, W, T = 8, 720, 1280, 12
    seqs = [synth.make_sequence(T, H, W, iso=3200, seed=6100 + b, device="cuda") for b in range(B)]
    raw = torch.stack([s.raw for s in seqs], 1)
    fl = torch.stack([s.flow_prev for s in seqs], 1)
    del seqs
    res = []
    for fused in (1, 0):
        rt = RvddRuntime("convunet+feat", 0, B, H, W, 0)
        rt.set_option("fuse_upsample", fused)
        rt.load_state_dict(sd)
        outs = torch.empty(T - 1, B, 3, H, W, device="cuda")
        for t in range(1, T):
            rt.step(raw[0] if t == 1 else None, raw[t], None, fl[t], None, out=outs[t - 1])
        res.append((outs, rt.get_state()[1].clone()))
        rt.close()
    (a, fa), (b, fb) = res
    if not torch.equal(a, b):
        d = (a - b).abs()
        bad = [(t, s, int((d[t, s] > 0).sum()), float(d[t, s].max())) for t in range(T - 1) for s in range(B) if bool((d[t, s] > 0).any())]
        raise AssertionError(f"fused and two-kernel upsample differ in {len(bad)} (frame, slot) pairs, first {bad[:4]}")
    assert torch.equal(fa, fb)


def test_c1_one_sequence_run_to_run():
    """BASELINE's C1 as it states it -- ONE 256x256 sequence -- takes launches no batch of eight takes (the one-kernel pre-stage, the
    output-channel split of the small conv launches): 40 repetitions of a 30-frame sequence through one runtime, all identical."""
    from rvdd_release_amd import synth
    from rvdd_release_amd.runtime import RvddRuntime
    sd = load_weights("recurrent-convunet-iso3200")
    H = W = 256
    T = 30
    s = synth.make_sequence(T, H, W, iso=3200, seed=6200, device="cuda")
    rt = RvddRuntime("convunet", 0, 1, H, W, 0)
    rt.load_state_dict(sd)

    def run():
        rt.reset()
        outs = torch.empty(T - 1, 1, 3, H, W, device="cuda")
        for t in range(1, T):
            rt.step(s.raw[0][None] if t == 1 else None, s.raw[t][None], None, s.flow_prev[t][None], None, out=outs[t - 1])
        return outs

    ref = run()
    for rep in range(40):
        assert torch.equal(run(), ref), rep
    rt.close()


def test_c5_share_720p_ninety_frames():
    """The per-GPU share of BASELINE config C5 at its real size: 8 sequences x 90 frames of 1280x720 in lockstep.
    * run to run: every one of the 8 x 89 output frames identical;
    * a 30-frame run of the same sequences in ANOTHER slot order gives, slot by slot, the first 29 frames of the
      90-frame run bit for bit (no state leaks between slots or across a reset, the position in the batch never matters);
    * slot 0 against the oracle at full size on frames 1 and 2, and on frame 89 as ONE oracle step from the runtime's
      own state after frame 88 (89 full-size oracle frames would take minutes; the whole-length drift is measured
      against the oracle in test_whole_sequence_vs_oracle[C5] at a size the CPU can follow)."""
    from rvdd_release_amd import synth
    from rvdd_release_amd.runtime import RvddRuntime
    sd = load_weights("recurrent-convunet+feat-iso3200")
    B, H, W, T = 8, 720, 1280, 90
    raws, flows, gts = [], [], {}
    for b in range(B):
        s = synth.make_sequence(T, H, W, iso=3200, seed=5000 + b, device="cuda")
        raws.append(s.raw)
        flows.append(s.flow_prev)
        if b == 0:
            gts = {t: s.gt[t].cpu() for t in (1, 2, T - 1)}
        del s
    raw = torch.stack(raws, 1)            # [T,B,4,h,w]
    fl = torch.stack(flows, 1)
    del raws, flows
    rt = RvddRuntime("convunet+feat", 0, B, H, W, 0)
    rt.load_state_dict(sd)

    def run(n_frames, order, keep_state_before=None):
        rt.reset()
        outs = torch.empty(n_frames - 1, B, 3, H, W, device="cuda")
        state = None
        r, f = raw[:, order], fl[:, order]
        for t in range(1, n_frames):
            if keep_state_before == t:
                state = rt.get_state()
            rt.step(r[t - 1] if t == 1 else None, r[t], None, f[t], None, out=outs[t - 1])
        return outs, state

    ident = list(range(B))
    a, state88 = run(T, ident, keep_state_before=T - 1)
    a2, _ = run(T, ident)
    if not torch.equal(a, a2):      # say where: the first frame and slot that differ, how many elements, how far
        d = (a - a2).abs()
        bad = [(t, b, int((d[t, b] > 0).sum()), float(d[t, b].max())) for t in range(d.shape[0]) for b in range(B) if bool((d[t, b] > 0).any())]
        ys = torch.nonzero(d[bad[0][0], bad[0][1]] > 0)
        raise AssertionError(f"two runs differ: {len(bad)} (frame, slot) pairs, first {bad[:4]}, rows {int(ys[:, 1].min())}-{int(ys[:, 1].max())}, "
                             f"columns {int(ys[:, 2].min())}-{int(ys[:, 2].max())}")
    del a2
    order = [(b + 3) % B for b in range(B)]
    short, _ = run(30, order)
    for slot, b in enumerate(order):
        assert torch.equal(short[:, slot], a[:29, b]), (slot, b)
    del short
    # ---- the oracle at full size
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    c = lambda x: x[None].cpu()
    orc = O.RecurrentOracle(sd, future=0)
    for t in (1, 2):
        want = orc.step(c(raw[t - 1, 0]), c(raw[t, 0]), None, c(fl[t, 0]), None, first=(t == 1))[0]
        got = a[t - 1, 0].cpu()
        assert (got - want).abs().max() < 1e-4 and parity_psnr(got, want) > 120.0, t
        assert abs(O.psnr(got[None], gts[t][None]) - O.psnr(want[None], gts[t][None])) < 0.01
    den88, feat88 = state88
    orc.lastden, orc.lastfeat = den88[0:1].cpu(), feat88[0:1].cpu()
    t = T - 1
    want = orc.step(c(raw[t - 1, 0]), c(raw[t, 0]), None, c(fl[t, 0]), None, first=False)[0]
    got = a[t - 1, 0].cpu()
    assert (got - want).abs().max() < 1e-4 and parity_psnr(got, want) > 120.0, float((got - want).abs().max())
    assert abs(O.psnr(got[None], gts[t][None]) - O.psnr(want[None], gts[t][None])) < 0.01
    assert torch.isfinite(a).all()
    rt.close()


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_rccl_collectives_run_at_world_size_one():
    """The driver's own launch form, `python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`, as a
    CHILD process: the process group must be initialised on RCCL (backend "nccl") although the world has one rank,
    `n_ranks_seen` must come out of an all_reduce on a DEVICE tensor, and the output collate must be an all-gather on
    device tensors -- the first RCCL init of this code must not be the one on the 8-GPU box."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)        # bench.py must set it itself when the launcher did not
    env["OMP_NUM_THREADS"] = "4"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "1", "--config", "C5", "--frames", "4",
           "--steps", "1", "--warmup", "1", "--collate-outputs", "--cpu-frames", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    d = line["distributed"]
    assert d["process_group"] is True and d["backend"] == "nccl", d
    assert d["collective_tensors_on"].startswith("cuda"), d
    assert d["HSA_ENABLE_IPC_MODE_LEGACY"] == "0", d
    assert line["n_ranks_seen"] == 1 and line["n_gpus"] == 1
    assert line["config"]["workload"].startswith("C5") and line["config"]["sequences_in_lockstep"] == 8
    col = line["collate"]
    assert col["gathered_shape"] == [1, 3, 8, 3, 720, 1280] and col["bytes_per_rank"] == 3 * 8 * 3 * 720 * 1280 * 4
    assert line["value"] > 0 and 20.0 < line["task_psnr_db"] < 60.0


@pytest.mark.parametrize("which", [[0, 1, 0, 1], [0, 1, 1, 0, 0, 1, 0, 1]], ids=["B4", "B8"])
def test_c4_batch_720p(which):
    """BASELINE config C4 at 1280x720, ConvNeXtUnet+feat+future through the fused ConvBlock kernel, at B = 4 and at the
    batch bench.py times (B = 8: the PROJ-epilogue and output-split forms are picked by launch size): position in the
    batch never matters (slots that hold the same sequence: equal bits), a sequence of the batch equals the same
    sequence alone bit for bit (a tile's arithmetic does not depend on how many tiles the launch has), and slot 0
    matches the full-size oracle on the first frame."""
    from rvdd_release_amd import synth
    from rvdd_release_amd.runtime import RvddRuntime
    stem = "recurrent-ConvNeXtUnet+feat-future-iso3200"
    sd = load_weights(stem)
    H, W, T = 720, 1280, 4
    seqs = [synth.make_sequence(T, H, W, iso=3200, seed=3100 + b, device="cuda") for b in range(2)]

    def run(which):
        rt = RvddRuntime("next+feat", 1, len(which), H, W, 0)
        rt.load_state_dict(sd)
        st = lambda f: torch.stack([f(seqs[b]) for b in which], 0)
        outs = [rt.step(st(lambda s: s.raw[t - 1]) if t == 1 else None, st(lambda s: s.raw[t]), st(lambda s: s.raw[t + 1]),
                        st(lambda s: s.flow_prev[t]), st(lambda s: s.flow_next[t])).clone() for t in (1, 2)]
        rt.close()
        return outs

    four = run(which)
    alone = run([1])
    first = {b: which.index(b) for b in (0, 1)}
    for t in range(2):
        for slot, b in enumerate(which):
            assert torch.equal(four[t][slot], four[t][first[b]]), (t, slot)
        assert torch.equal(four[t][first[1]], alone[t][0])
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    c = lambda x: x[None].cpu()
    s0 = seqs[0]
    want = O.RecurrentOracle(sd, future=1).step(c(s0.raw[0]), c(s0.raw[1]), c(s0.raw[2]), c(s0.flow_prev[1]), c(s0.flow_next[1]), first=True)[0]
    got = four[0][0].cpu()
    assert (got - want).abs().max() < 1e-4 and parity_psnr(got, want) > 120.0, float((got - want).abs().max())
