"""Host-side logic that needs no GPU: registries, option handling, weight key
tables, the synthetic generator."""
import os

import pytest
import torch

from conftest import VARIANTS, WEIGHTS, load_weights


def test_parse_kwargs_matches_reference_minilanguage():
    from rvdd_release_amd.networks import parse_kwargs
    assert parse_kwargs("convunet-mode=fixedfeatures+feat") == {"mode": "fixedfeatures+feat"}
    assert parse_kwargs("newunet-mode=feat-filters=48-foo=none-bar=true") == {
        "mode": "feat", "filters": 48, "foo": None, "bar": True}
    assert parse_kwargs("newunet") == {}


def test_define_net_arch_registry():
    from rvdd_release_amd import networks as N
    assert isinstance(N.define_net_arch(6, 3, "convunet-mode=fixedfeatures", gpu_ids=[0]), N.UNet_FixedFeatures)
    f = N.define_net_arch(9, 3, "convunet-mode=fixedfeatures+feat", gpu_ids=[0])
    assert isinstance(f, N.UNet_FixedFeatures_feat) and f.NoPF == 1 and f.future == 1
    assert isinstance(N.define_net_arch(6, 3, "newunet", gpu_ids=[0]), N.NewUNet)
    assert isinstance(N.define_net_arch(9, 3, "newunet-mode=feat", gpu_ids=[0]), N.NewUNet_feat)
    with pytest.raises(NotImplementedError):
        N.define_net_arch(6, 3, "resnet_9blocks")
    with pytest.raises(NotImplementedError):
        N.define_net_arch(6, 3, "convunet")                       # growing-filter U-Net: no checkpoint
    with pytest.raises(NotImplementedError):
        N.define_net_arch(6, 3, "convunet-mode=fixedfeatures-filters=64")
    with pytest.raises(Exception, match="does not exist"):
        N.define_net_arch(6, 3, "convunet-mode=nonsense")
    with pytest.raises(NotImplementedError):
        N.define_net_arch(12, 3, "newunet")


def test_feat_net_surface_without_gpu():
    from rvdd_release_amd import networks as N
    net = N.define_net_arch(6, 3, "convunet-mode=fixedfeatures+feat", gpu_ids=[0])
    assert net.get_current_features() == [None]
    with pytest.raises(Exception, match="Old features is None"):
        net(torch.zeros(1, 6, 16, 16))
    t = torch.zeros(1, 48, 16, 16)
    net.set_rec_features([t])
    assert net.get_current_features()[0] is t
    sd = load_weights("recurrent-convunet+feat-iso3200")
    net.load_state_dict(sd)
    assert list(net.state_dict()) == list(sd)
    assert sum(p.numel() for p in net.parameters()) == 563763
    assert net.eval() is net
    with pytest.raises(NotImplementedError):
        net.train()


def test_model_registry_and_options():
    from rvdd_release_amd.models import create_model, find_model_using_name
    from rvdd_release_amd.models.recurrent_model import recurrentModel
    from rvdd_release_amd.options import make_opt
    assert find_model_using_name("recurrent") is recurrentModel
    with pytest.raises(ModuleNotFoundError):
        find_model_using_name("cyclegan")
    opt = make_opt(netDenoiser="convunet-mode=fixedfeatures+feat", feature_rec=True)
    assert opt.name == "recurrent-convunet-mode=fixedfeatures+feat-warp-i3o3"   # base_options.py:131-136
    m = create_model(opt)
    assert m.loss_names == ['L1', 'PSNR', 'Denoiser'] and m.visual_names == ['denoised']
    assert m.training_unrollings == 4 and m.get_current_losses() == {'L1': 0, 'PSNR': 0, 'Denoiser': 0}
    assert m.optimizers[0].param_groups[0]['lr'] == opt.lr
    with pytest.raises(AttributeError):
        make_opt(no_such_flag=1)
    assert create_model(make_opt(no_warp=True)).opt.no_warp                  # --no_warp is built (tests/test_gpu_parity.py)
    for flag in ("no_predemosaic", "raw_gt"):
        with pytest.raises(NotImplementedError):
            create_model(make_opt(**{flag: True}))
    with pytest.raises(ValueError):
        create_model(make_opt(netDenoiser="convunet-mode=fixedfeatures+feat"))   # needs --feature_rec
    with pytest.raises(RuntimeError, match="no CPU path"):
        create_model(make_opt(gpu_ids=[]))
    with pytest.raises(NotImplementedError):
        m.optimize_parameters()


def test_load_networks_reads_pth_and_safetensors(tmp_path):
    from rvdd_release_amd.models import create_model
    from rvdd_release_amd.options import make_opt
    sd = load_weights("recurrent-convunet-iso3200")
    torch.save(sd, tmp_path / "ep7_net_Denoise.pth")
    for stem in (str(tmp_path / "ep7"), os.path.join(WEIGHTS, "recurrent-convunet-iso3200")):
        opt = make_opt(path2epoch=stem)
        m = create_model(opt)
        m.setup(opt)
        assert list(m.netDenoise.state_dict()) == list(sd)
    with pytest.raises(FileNotFoundError):
        opt = make_opt(path2epoch=str(tmp_path / "missing"))
        create_model(opt).setup(opt)


def test_every_shipped_checkpoint_matches_its_key_table():
    """The key/shape table the runtime enforces (runtime.hip expected_keys) is the
    reference's: count tensors and parameters."""
    fam = {"convunet": (48, 522243), "convunet-future": (48, 523539), "convunet+feat": (50, 563763),
           "convunet+feat-future": (50, 565059), "ConvNeXtUnet": (226, 523635), "ConvNeXtUnet+feat-future": (237, 549651)}
    stems = []
    for iso in ("iso3200", "iso12800"):
        for f in ("convunet", "convunet-future", "convunet+feat", "convunet+feat-future", "ConvNeXtUnet",
                  "ConvNeXtUnet+feat-future"):
            stems.append((f"recurrent-{f}-{iso}", fam[f]))
        for f, k in (("convunet", "convunet"), ("convunet-future", "convunet-future"), ("convunet-no_warp", "convunet"),
                     ("convunet-no_warp-future", "convunet-future")):
            stems.append((f"non_recurrent-{f}-{iso}", fam[k]))
    assert len(stems) == 20                                   # every file of the reference's trained-nets/
    import glob, os
    from conftest import WEIGHTS
    assert sorted(os.path.basename(p)[:-12] for p in glob.glob(os.path.join(WEIGHTS, "*.safetensors"))) == sorted(s for s, _ in stems)
    for stem, (nt, npar) in stems:
        sd = load_weights(stem)
        assert len(sd) == nt and sum(v.numel() for v in sd.values()) == npar, stem
        assert all(v.dtype == torch.float32 for v in sd.values())


def test_synth_is_deterministic_and_shaped():
    from rvdd_release_amd import synth
    a = synth.make_sequence(4, 32, 48, iso=3200, seed=11)
    b = synth.make_sequence(4, 32, 48, iso=3200, seed=11)
    c = synth.make_sequence(4, 32, 48, iso=12800, seed=12)
    assert a.raw.shape == (4, 4, 16, 24) and a.flow_prev.shape == (4, 2, 16, 24) and a.gt.shape == (4, 3, 32, 48)
    assert torch.equal(a.raw, b.raw) and torch.equal(a.gt, b.gt) and not torch.equal(a.raw, c.raw)
    assert a.gt.min() >= -1 and a.gt.max() <= 1
    # flow convention: cur(x) ~ prev(x + flow): the next-flow is the mirror of the prev-flow
    assert torch.allclose(a.flow_prev, -a.flow_next)
    # the GBRG mosaic of the clean frame is what the noise was added to: residual ~ noise model
    g = a.gt[:, 1, 0::2, 0::2]
    assert (a.raw[:, 0] - g).abs().mean() < 0.05


def test_bench_traffic_is_stamped_with_the_kernel_sources_it_was_measured_on():
    """profiles/traffic.json (the PMC bytes bench.py quotes in roofline.traffic) carries the hash of csrc/ it was measured on;
    bench.py withholds the number -- and says so -- for any other sources.  The committed tree must be consistent: a kernel
    change is followed by tools/r06_evidence.sh (tools/gpu_profile.sh + tools/pmc_summary.py) before it is committed."""
    import json
    import os
    import bench
    tj = json.load(open(os.path.join(bench.REPO, "profiles", "traffic.json")))
    now = bench.csrc_hash()
    for cfg in ("C2", "C4"):
        assert tj[cfg].get("csrc_sha256_16") == now, (cfg, tj[cfg].get("csrc_sha256_16"), now)
    k = {"launches": 10, "avg_us": 222.0, "tflops": 370.0, "gbps": 3400.0, "bytes_per_launch": 768e6}
    r = bench.roofline_of("conv3x3h_kernel<48, 1, false, false>", k, "C2", "convunet+feat", "test")
    assert r["traffic"] and "STALE" not in r["traffic_source"]
    assert r["peak_sustained"] == bench.F16_SUSTAINED_TFLOPS and abs(r["frac_executed_sustained"] - 370.0 * 3 * 448 / 432 / 1926.0) < 1e-3
    # other sources: withheld
    real = bench.csrc_hash
    try:
        bench.csrc_hash = lambda: "0" * 16
        r2 = bench.roofline_of("conv3x3h_kernel<48, 1, false, false>", k, "C2", "convunet+feat", "test")
    finally:
        bench.csrc_hash = real
    assert r2["traffic"] is None and "STALE" in r2["traffic_source"]


def test_gpu_sampler_window_and_failure_modes():
    """bench.GpuSampler never raises, reports only the samples inside the asked window, and summarises clock and power."""
    import time
    import bench
    s = bench.GpuSampler.__new__(bench.GpuSampler)
    s.samples = [(1.0, 1900, 1300.0), (2.0, 1950, 1350.0), (3.0, None, None), (9.0, 2400, 300.0)]
    s.error = None
    w = s.window(0.5, 3.5)
    assert w["samples"] == 3 and w["sclk_mhz_mean"] == 1925.0 and w["sclk_mhz_min"] == 1900 and w["package_w_max"] == 1350.0
    assert s.window(4.0, 5.0)["samples"] == 0
    live = bench.GpuSampler(0, 0.05)          # no GPU here: rocm-smi prints nothing useful or is absent -- no exception either way
    time.sleep(0.3)
    live.close()
    assert isinstance(live.window(0.0, time.perf_counter()), dict)
