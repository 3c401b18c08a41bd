#!/usr/bin/env python3
"""Golden sequences for the `--no_warp` checkpoints (scripts/test-non_recurrent-no_warp-*.sh of the reference),
captured from the reference itself like tools/make_golden.py does for the warping variants (build container only):

    python3 tools/make_golden_nowarp.py

Writes weights/non_recurrent-convunet-no_warp[-future]-iso3200.safetensors and tests/golden/seq_nowarp*.npz."""
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import make_golden as MG  # noqa: E402

# name: (netDenoiser, future, checkpoint stem, extra flags, feature_rec)
CASES = {"nowarp-iso3200": ("convunet-mode=fixedfeatures", 0, "non_recurrent-convunet-no_warp-iso3200", ["--no_warp"], False),
         "nowarp-future-iso3200": ("convunet-mode=fixedfeatures", 1, "non_recurrent-convunet-no_warp-future-iso3200", ["--no_warp"], False),
         # --prev_noisy_frame (recurrent_model.py:33, :335-337): no checkpoint was trained with it; any one runs with it
         # --warp_raw (:149-152): likewise, no checkpoint of its own
         "warpraw-iso3200": ("convunet-mode=fixedfeatures", 0, "recurrent-convunet-iso3200", ["--warp_raw"], False),
         "warpraw-future-iso3200": ("convunet-mode=fixedfeatures", 1, "recurrent-convunet-future-iso3200", ["--warp_raw"], False),
         "prevnoisy-feat-iso3200": ("convunet-mode=fixedfeatures+feat", 0, "recurrent-convunet+feat-iso3200", ["--prev_noisy_frame", "--feature_rec"], True)}


def main(argv=None):
    args = MG.parse_args(argv)
    gold = os.path.abspath(args.out) if args.out else MG.GOLD
    wdir = os.path.join(gold, "weights") if args.out else MG.WDIR
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    MG._install_standins()
    sys.path.insert(0, MG.REF)
    os.makedirs(gold, exist_ok=True)
    os.makedirs(wdir, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="rvdd_golden_")
    os.chdir(tmp)
    torch.manual_seed(4321)
    torch.set_num_threads(8)
    from safetensors.torch import save_file
    synth = MG.load_synth()
    import models, networks, options.train_options, util.flow_utils
    MG.assert_reference_modules(models, networks, options.train_options, util.flow_utils)
    from options.train_options import TrainOptions
    from models import create_model
    for name, (netstr, fut, stem, extra, feat) in CASES.items():
        sys.argv = ["x", "--gpu_ids", "-1", "--netDenoiser", netstr, "--path2epoch", os.path.join(MG.REF, "trained-nets", stem),
                    "--checkpoints_dir", tmp] + extra + (["--future_patch_depth", "1"] if fut else [])
        opt = TrainOptions().parse()
        model = create_model(opt)
        model.setup(opt)
        opt.isTrain = model.isTrain = False
        model.eval()
        sd = {k: v.detach().clone().contiguous() for k, v in model._netDenoise.state_dict().items()}
        if "--no_warp" in extra:
            save_file(sd, os.path.join(wdir, stem + ".safetensors"),
                      metadata={"netDenoiser": netstr, "feature_rec": "0", "future_patch_depth": str(fut), "no_warp": "1",
                                "source": stem + "_net_Denoise.pth"})
        no_warp = "--no_warp" in extra
        T, H, W = 6, 32, 48
        seq = synth.make_sequence(T, H, W, iso=3200, seed=2000 + len(name))
        outs, l1s, psnrs = [], [], []
        for t in range(1, T - fut):
            frames = [seq.raw[t - 1], seq.raw[t]] + ([seq.raw[t + 1]] if fut else [])
            flows = [] if no_warp else torch.stack([seq.flow_prev[t]] + ([seq.flow_next[t]] if fut else []), 0)[None]
            data = {"n": torch.cat(frames, 0)[None], "flow": flows,                    # the dataset yields [] with --no_warp
                    "gt": torch.cat((seq.gt[t - 1], seq.gt[t]), 0)[None],
                    "n_path": [f"seq/{t:03d}.tif"], "gt_path": [f"seq/{t:03d}.tif"], "FirstOfVideo": t == 1}
            model.set_input(data)
            model.test()
            model.compute_losses()
            losses = model.get_current_losses()
            outs.append(model.denoised[0].numpy().copy())
            l1s.append(losses["L1"])
            psnrs.append(losses["PSNR"])
        np.savez(os.path.join(gold, f"seq_{name}.npz"), raw=seq.raw.numpy(), gt=seq.gt.numpy(), denoised=np.stack(outs, 0),
                 flow_prev=seq.flow_prev.numpy(), flow_next=seq.flow_next.numpy(),
                 L1=np.array(l1s, np.float64), PSNR=np.array(psnrs, np.float64))
        print(f"[golden] {name}: {len(outs)} frames, PSNR {psnrs}")


if __name__ == "__main__":
    main()
