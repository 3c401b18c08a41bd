#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ and weights/ from the
REFERENCE ITSELF, imported read-only from /root/reference in the build
container (recipe: SURVEY.md section 8c).  Run once per change of the fixture
set:

    PYTHONDONTWRITEBYTECODE=1 python3 tools/make_golden.py

The reference never travels to the GPU box; only the small data files written
here (inputs + expected outputs) and this script are committed.

Third-party packages the reference imports but that are absent from this image
(opt_einsum, cv2, skimage, iio, torchvision) get import-time stand-ins in
``sys.modules`` -- reference files are untouched.  Only the ``opt_einsum``
stand-in touches hot-path arithmetic: ``contract("c, b c ... -> b c ...")`` is
a per-channel multiply (networks/new_unet.py:28,44), mapped to torch.einsum.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
WDIR = os.path.join(REPO, "weights")


def _install_standins():
    oe = types.ModuleType("opt_einsum")
    oe.contract = lambda expr, *ops, backend=None: torch.einsum(expr.replace(" ", ""), *ops)
    sys.modules["opt_einsum"] = oe
    sys.modules["cv2"] = types.ModuleType("cv2")
    sk = types.ModuleType("skimage")
    skio = types.ModuleType("skimage.io")
    skc = types.ModuleType("skimage.color")
    skc.rgb2gray = lambda x: x
    sk.io, sk.color = skio, skc
    sys.modules.update({"skimage": sk, "skimage.io": skio, "skimage.color": skc})
    iio = types.ModuleType("iio")
    iio.read = lambda p: (_ for _ in ()).throw(RuntimeError("iio stand-in"))
    iio.write = lambda p, x: None
    sys.modules["iio"] = iio
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class _Lambda:
        def __init__(self, f): self.f = f
        def __call__(self, x): return self.f(x)

    class _Compose:
        def __init__(self, ts): self.ts = ts
        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class _ToTensor:
        def __call__(self, x): return torch.from_numpy(np.ascontiguousarray(x)).permute(2, 0, 1)

    tvt.Lambda, tvt.Compose, tvt.ToTensor = _Lambda, _Compose, _ToTensor
    tv.transforms = tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt})


VARIANTS = {
    # name: (netDenoiser string, feature_rec, future, checkpoint stem)
    "basic-iso3200": ("convunet-mode=fixedfeatures", False, 0, "recurrent-convunet-iso3200"),
    "basic-future-iso3200": ("convunet-mode=fixedfeatures", False, 1,
                             "recurrent-convunet-future-iso3200"),
    "feat-iso3200": ("convunet-mode=fixedfeatures+feat", True, 0, "recurrent-convunet+feat-iso3200"),
    "feat-future-iso12800": ("convunet-mode=fixedfeatures+feat", True, 1,
                             "recurrent-convunet+feat-future-iso12800"),
    "next-iso3200": ("newunet", False, 0, "recurrent-ConvNeXtUnet-iso3200"),
    "next-feat-future-iso3200": ("newunet-mode=feat", True, 1,
                                 "recurrent-ConvNeXtUnet+feat-future-iso3200"),
}


def build_reference_model(name, tmp):
    """Exactly validate.py:117-138 minus the dataset (SURVEY.md section 8c recipe)."""
    net, feat, fut, stem = VARIANTS[name]
    argv = ["x", "--gpu_ids", "-1", "--netDenoiser", net, "--path2epoch",
            os.path.join(REF, "trained-nets", stem), "--checkpoints_dir", tmp]
    if feat:
        argv.append("--feature_rec")
    if fut:
        argv += ["--future_patch_depth", str(fut)]
    sys.argv = argv
    from options.train_options import TrainOptions
    from models import create_model
    opt = TrainOptions().parse()
    model = create_model(opt)
    model.setup(opt)
    opt.isTrain = False
    model.isTrain = False
    model.eval()
    return model, opt


def load_synth():
    """The input generator of this repo, loaded BY FILE: its package mirrors the reference's module names
    (util/, models/, networks/, options/, data/), so the package directory must never be on sys.path here."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("rvdd_synth", os.path.join(REPO, "rvdd-release_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    sys.modules["rvdd_synth"] = synth
    spec.loader.exec_module(synth)
    return synth


def assert_reference_modules(*mods):
    """Every module the fixtures are computed with must come from the reference tree, not from this repo's
    name-for-name plugin surface (a shadowed import would write 'golden' vectors from the build's own kernels)."""
    for m in mods:
        f = os.path.realpath(getattr(m, "__file__", "") or "")
        if not f.startswith(os.path.realpath(REF) + os.sep):
            raise RuntimeError(f"{m.__name__} was imported from {f!r}, not from {REF}: refusing to write fixtures")


def parse_args(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default=None, help="write fixtures here instead of tests/golden (weights go to <out>/weights)")
    ap.add_argument("--only", default=None, help="comma-separated subset of fixture names (ops, or variant / long names)")
    return ap.parse_args(argv)


def main(argv=None):
    global GOLD, WDIR
    args = parse_args(argv)
    if args.out:
        GOLD, WDIR = os.path.abspath(args.out), os.path.join(os.path.abspath(args.out), "weights")
    only = set(args.only.split(",")) if args.only else None
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    _install_standins()
    sys.path.insert(0, REF)
    os.makedirs(GOLD, exist_ok=True)
    os.makedirs(WDIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="rvdd_golden_")
    os.chdir(tmp)
    torch.manual_seed(1234)
    torch.set_num_threads(8)

    from safetensors.torch import save_file
    import util.Hamilton_Adam_demo, util.flow_utils, models, networks, options.train_options
    assert_reference_modules(util.Hamilton_Adam_demo, util.flow_utils, models, networks, options.train_options)
    from util.Hamilton_Adam_demo import HamiltonAdam
    from util.flow_utils import warp, upsample_factor_2
    synth = load_synth()

    # ---- per-op fixtures ------------------------------------------------
    ha = HamiltonAdam("gbrg")
    raw = torch.rand(2, 8, 18, 26) * 2 - 1
    # plateaus so that the sign() selections also see exact ties
    raw[0, :4, 4:9, 5:12] = 0.25
    raw[1, 4:, :, :6] = -0.5
    np.savez(os.path.join(GOLD, "op_hamilton_adams.npz"), raw=raw.numpy(),
             rgb=ha(raw).numpy(), remosaick=ha.remosaick(ha(raw)[:, :3]).numpy())

    x = torch.randn(2, 5, 37, 53)
    fl = torch.randn(2, 2, 37, 53) * 3.0
    fl[0, :, :6, :] *= 12.0          # far out of range on every side
    fl[1, :, :, -5:] += 40.0
    fl[1, :, -4:, :] -= 40.0
    fl[0, :, 10:14, 10:14] = 0.0     # exact integer coordinates
    fl[0, 0, 20:24, 20:24] = 0.5
    y, mask = warp(x, fl, interp="bicubic")
    np.savez(os.path.join(GOLD, "op_warp_bicubic.npz"), x=x.numpy(), flow=fl.numpy(),
             y=y.numpy(), mask=mask.numpy())

    f = torch.randn(1, 1, 2, 2, 18, 26)
    np.savez(os.path.join(GOLD, "op_upsample_flow.npz"), flow=f.numpy(),
             up=upsample_factor_2(f, multiply_by=2).numpy())

    # ---- per-variant: weights, single forwards, short sequences ----------
    for name, (netstr, feat, fut, stem) in VARIANTS.items():
        model, opt = build_reference_model(name, tmp)
        net = model._netDenoise
        sd = {k: v.detach().clone().contiguous() for k, v in net.state_dict().items()}
        save_file(sd, os.path.join(WDIR, stem + ".safetensors"),
                  metadata={"netDenoiser": netstr, "feature_rec": str(int(feat)),
                            "future_patch_depth": str(fut), "source": stem + "_net_Denoise.pth"})
        cin = 3 * (2 + fut)

        fwd = {}
        for (H, W) in ((20, 28), (16, 24)):
            xin = torch.randn(1, cin, H, W) * 0.5
            tag = f"{H}x{W}"
            fwd[f"x_{tag}"] = xin.numpy()
            with torch.no_grad():
                if feat:
                    fin = torch.relu(torch.randn(1, 48, H, W)) * 0.3
                    net.set_rec_features([fin])
                    out = net(xin)
                    fwd[f"feat_in_{tag}"] = fin.numpy()
                    fwd[f"feat_out_{tag}"] = net.get_current_features()[0].numpy()
                else:
                    out = net(xin)
            fwd[f"out_{tag}"] = out.numpy()
        np.savez(os.path.join(GOLD, f"net_{name}.npz"), **fwd)

        # a short sequence driven exactly like validate.py:75-88
        T, H, W = 6, 32, 48
        iso = 12800 if "12800" in name else 3200
        seq = synth.make_sequence(T, H, W, iso=iso, seed=1000 + len(name))
        outs, l1s, psnrs = [], [], []
        for t in range(1, T - fut):
            frames = [seq.raw[t - 1], seq.raw[t]] + ([seq.raw[t + 1]] if fut else [])
            flows = [seq.flow_prev[t]] + ([seq.flow_next[t]] if fut else [])
            data = {"n": torch.cat(frames, 0)[None], "flow": torch.stack(flows, 0)[None],
                    "gt": torch.cat((seq.gt[t - 1], seq.gt[t]), 0)[None],
                    "n_path": [f"seq/{t:03d}.tif"], "gt_path": [f"seq/{t:03d}.tif"],
                    "FirstOfVideo": t == 1}
            model.set_input(data)
            model.test()
            model.compute_losses()
            losses = model.get_current_losses()
            outs.append(model.denoised[0].numpy().copy())
            l1s.append(losses["L1"])
            psnrs.append(losses["PSNR"])
        d = dict(raw=seq.raw.numpy(), flow_prev=seq.flow_prev.numpy(),
                 flow_next=seq.flow_next.numpy(), gt=seq.gt.numpy(),
                 denoised=np.stack(outs, 0), L1=np.array(l1s, np.float64),
                 PSNR=np.array(psnrs, np.float64))
        if feat:
            d["feat_last"] = net.get_current_features()[0][0].numpy()
        np.savez(os.path.join(GOLD, f"seq_{name}.npz"), **d)
        print(f"[golden] {name}: {len(outs)} frames, PSNR {psnrs}")


if __name__ == "__main__":
    main()
