#!/usr/bin/env python3
"""Whole-sequence golden fixtures: the REFERENCE ITSELF driven over the sequence
lengths BASELINE.json quotes (30 frames for C2 / C3 / C4, 90 for C5), imported
read-only from /root/reference in the build container exactly like
tools/make_golden.py (same import-time stand-ins, same call sequence as
validate.py:64-88; models/recurrent_model.py:335-345 feeds every output back).

    PYTHONDONTWRITEBYTECODE=1 python3 tools/make_golden_long.py

The short fixtures stop after 5 outputs; these pin the recurrence over its whole
length -- error growth through 29 / 28 / 89 dependent steps would show here.
To keep the files small they hold, per sequence,
  * the generator arguments of the inputs (rvdd-release_amd/synth.py is
    deterministic; every 7th raw / flow / gt frame is stored so that a test can
    verify it regenerated the same inputs),
  * the reference's output at a few frames spread over the sequence, the last one
    included, and its recurrent features after the last frame,
  * the reference's L1 and PSNR of EVERY frame (recurrent_model.py:512-525).
"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402

# name -> (variant of make_golden.VARIANTS, BASELINE config it mirrors, frames, H, W, seed)
LONG = {
    "long30-feat-iso3200": ("feat-iso3200", "C2", 30, 32, 48, 7002),
    "long30-feat-future-iso12800": ("feat-future-iso12800", "C3", 30, 32, 48, 7003),
    "long30-next-feat-future-iso3200": ("next-feat-future-iso3200", "C4", 30, 32, 48, 7004),
    "long90-feat-iso3200": ("feat-iso3200", "C5", 90, 32, 48, 7005),
}
CHECK_EVERY = 7


def keep_frames(n_out):
    """Output indices whose frames are stored: first two, every tenth, the last two."""
    return sorted(set([0, 1, n_out - 2, n_out - 1] + list(range(9, n_out, 10))))


def main(argv=None):
    args = MG.parse_args(argv)
    gold = os.path.abspath(args.out) if args.out else MG.GOLD
    only = set(args.only.split(",")) if args.only else None
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    MG._install_standins()
    sys.path.insert(0, MG.REF)
    os.makedirs(gold, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="rvdd_golden_long_")
    os.chdir(tmp)
    torch.set_num_threads(8)
    import models, networks, options.train_options, util.flow_utils
    MG.assert_reference_modules(models, networks, options.train_options, util.flow_utils)
    synth = MG.load_synth()

    for name, (variant, cfg, T, H, W, seed) in LONG.items():
        if only and name not in only:
            continue
        _, feat, fut, _ = MG.VARIANTS[variant]
        iso = 12800 if "12800" in variant else 3200
        model, _ = MG.build_reference_model(variant, tmp)
        seq = synth.make_sequence(T, H, W, iso=iso, seed=seed)
        n_out = T - 1 - fut
        keep = keep_frames(n_out)
        outs, l1s, psnrs = {}, [], []
        for t in range(1, T - fut):
            frames = [seq.raw[t - 1], seq.raw[t]] + ([seq.raw[t + 1]] if fut else [])
            flows = [seq.flow_prev[t]] + ([seq.flow_next[t]] if fut else [])
            data = {"n": torch.cat(frames, 0)[None], "flow": torch.stack(flows, 0)[None],
                    "gt": torch.cat((seq.gt[t - 1], seq.gt[t]), 0)[None],
                    "n_path": [f"seq/{t:03d}.tif"], "gt_path": [f"seq/{t:03d}.tif"], "FirstOfVideo": t == 1}
            model.set_input(data)
            model.test()
            model.compute_losses()
            losses = model.get_current_losses()
            if t - 1 in keep:
                outs[t - 1] = model.denoised[0].numpy().copy()
            l1s.append(losses["L1"])
            psnrs.append(losses["PSNR"])
        d = dict(args=np.array([T, H, W, iso, seed, fut], np.int64), keep=np.array(keep, np.int64),
                 denoised=np.stack([outs[k] for k in keep], 0), L1=np.array(l1s, np.float64),
                 PSNR=np.array(psnrs, np.float64),
                 raw_check=seq.raw[::CHECK_EVERY].numpy(), flow_prev_check=seq.flow_prev[::CHECK_EVERY].numpy(),
                 flow_next_check=seq.flow_next[::CHECK_EVERY].numpy(), gt_check=seq.gt[::CHECK_EVERY].numpy())
        if feat:
            d["feat_last"] = model._netDenoise.get_current_features()[0][0].numpy()
        np.savez_compressed(os.path.join(gold, f"seq_{name}.npz"), **d)
        print(f"[golden-long] {name} ({cfg}): {n_out} frames, kept {keep}, PSNR first/last {psnrs[0]:.3f}/{psnrs[-1]:.3f}")


if __name__ == "__main__":
    main()
