#!/usr/bin/env python3
"""Convert every shipped checkpoint (trained-nets/*.pth: plain fp32 state_dicts, data under the reference's BSD-2
licence) to weights/<stem>.safetensors.  No reference code is imported: torch.load(weights_only=True) reads tensors
only.  Run in the build container (the GPU box has no /root/reference):  python tools/convert_checkpoints.py"""
import glob
import os
import sys

import torch
from safetensors.torch import load_file, save_file

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/trained-nets"


def meta(stem):
    nxt = "ConvNeXtUnet" in stem
    feat = "+feat" in stem
    fut = "-future" in stem
    net = ("newunet" + ("-mode=feat" if feat else "")) if nxt else ("convunet-mode=fixedfeatures" + ("+feat" if feat else ""))
    return {"netDenoiser": net, "feature_rec": str(int(feat)), "future_patch_depth": str(int(fut)),
            "no_warp": str(int("no_warp" in stem)), "source": stem + "_net_Denoise.pth"}


def main():
    n = 0
    for path in sorted(glob.glob(os.path.join(SRC, "*_net_Denoise.pth"))):
        stem = os.path.basename(path)[:-len("_net_Denoise.pth")]
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = {(k[7:] if k.startswith("module.") else k): v.detach().to(torch.float32).contiguous() for k, v in sd.items()}
        dst = os.path.join(REPO, "weights", stem + ".safetensors")
        if os.path.exists(dst):
            old = load_file(dst)
            assert set(old) == set(sd) and all(torch.equal(old[k], sd[k]) for k in sd), f"{stem}: differs from the committed file"
            print(f"= {stem}: {len(sd)} tensors, identical to the committed file")
            continue
        save_file(sd, dst, metadata=meta(stem))
        print(f"+ {stem}: {len(sd)} tensors, {sum(v.numel() for v in sd.values())} parameters")
        n += 1
    print(f"{n} new file(s)")


if __name__ == "__main__":
    main()
