"""The validation loop of the reference (validate.py:16-114) on the HIP runtime, minus file I/O:
`compute_validation(model, val_dataset, opt)` drives `set_input` / `test` / `compute_losses` per
output frame, latches `FirstOfVideo` on a change of video folder, and -- with
`opt.val_flow_from_denoised` -- recomputes the optical flow online from the previous DENOISED frame
(`compute_flows_from_denoised`, validate.py:16-38): remosaick -> TV-L1 -> backward flow, all on the
device (`rvdd_tvl1flow`).

`val_dataset` is any iterable of the dicts the reference's `infer4recDataset` yields
(data/infer4rec_dataset.py:226-230): 'n' [1,(2+f)*4,h,w], 'gt' [1,6,H,W], 'flow' [1,1+f,2,h,w],
'n_path', 'gt_path'.
"""
from __future__ import annotations

from os.path import basename, dirname, join
from typing import Callable, Dict, Iterable, Optional

import torch

from .util._ops import ops_runtime
from .util.Hamilton_Adam_demo import HamiltonAdam


def compute_flows_from_denoised(data: dict, model, opt) -> None:
    """validate.py:16-38.  The flow goes from the last noisy frame of `data['n']` to the
    remosaicked previous output.  The released reference cannot actually run this branch (it hands
    `remosaick` a squeezed 3-D tensor, validate.py:29-31 vs util/Hamilton_Adam_demo.py:237-238, and
    appends the same flow `opt.patch_depth - 1` times); this is its evident intent: ONE flow, shaped
    like the dataset's `data['flow']` ([1,1,2,h,w])."""
    if getattr(opt, "future_patch_depth", 0):
        raise NotImplementedError("rvdd: --val_flow_from_denoised with a future frame is not built "
                                  "(the reference pairs the NEXT frame with the previous output there)")
    dev = model.device
    ha = HamiltonAdam('gbrg')
    img2 = data['n'][0, -4:, :, :].to(dev, torch.float32)                  # last noisy frame, packed raw
    img1 = ha.remosaick(model.denoised.to(dev))[0]                          # previous output, re-mosaicked
    # singleiT = (x+1)/2 as an [h,w,c] image (library.py:67-68); CPPbridge takes the channel mean
    g2 = ((img2 + 1.0) / 2.0).mean(dim=0).contiguous()
    g1 = ((img1 + 1.0) / 2.0).mean(dim=0).contiguous()
    flow = ops_runtime(dev.index or 0).tvl1flow(g2, g1)                     # TVL1_flow(img2, img1): img1(x+u) ~ img2(x)
    data['flow'] = flow[None, None]


def init_validation_dataloader(opt):
    """validate.py:40-52."""
    import copy
    from .data import create_dataset
    opt_val = copy.deepcopy(opt)
    opt_val.dataroot = opt.val_dataroot
    opt_val.dataset_mode = opt.val_dataset_mode
    opt_val.max_dataset_size = float("inf")
    opt_val.videos = opt.val_videos
    opt_val.num_threads = 0
    opt_val.batch_size = 1
    opt_val.serial_batches = True
    if hasattr(opt, 'model_patch_depth'):
        opt_val.patch_depth = opt.model_patch_depth
    return create_dataset(opt_val)


def compute_validation(model, val_dataset: Iterable[Dict], opt, on_frame: Optional[Callable] = None,
                       val_image_dir: Optional[str] = None) -> dict:
    """validate.py:54-114: returns {'L1_valLoss', 'PSNR_valLoss', 'Denoiser_valLoss', 'lr'}.
    With `val_image_dir` (and a dataset made by `create_dataset`) every frame is written to
    <val_image_dir>/<video>/<frame>_denoised.tif and its losses appended to output.log, as the
    reference does; `on_frame(i, data, visuals, losses)` is an in-memory hook beside that."""
    val_flow_from_denoised = False if model.isTrain else getattr(opt, "val_flow_from_denoised", False)
    bak_isTrain = model.isTrain
    model.isTrain = False
    model.eval()
    val_losses = model.get_current_losses().copy()
    for k in val_losses:
        val_losses[k] = 0.0
    count = 0
    with torch.no_grad():
        lastvideopath = ''
        for i, data in enumerate(val_dataset):
            thisvideopath = dirname(data['gt_path'][0])
            data['FirstOfVideo'] = not thisvideopath == lastvideopath
            if (not opt.no_warp) and val_flow_from_denoised and not data['FirstOfVideo']:
                compute_flows_from_denoised(data, model, opt)
            model.set_input(data)
            model.test()
            model.compute_losses()
            losses = model.get_current_losses()
            if on_frame is not None:
                on_frame(i, data, model.get_current_visuals(), losses)
            if val_image_dir is not None:
                from .library import pathdiff, print_dict
                from .util.visualizer import save_images
                img_path = model.get_image_paths()
                if i % 40 == 0:
                    print('processing (%04d)-th image... %s' % (i, img_path))
                sfolder = pathdiff(img_path[0], val_dataset.dataset.n_paths)
                save_images(val_image_dir, model.get_current_visuals(), [basename(img_path[0])], subfolder=sfolder)
                print_dict(losses, suffix="", savefile=join(val_image_dir, "output.log"))
            lastvideopath = thisvideopath
            for k, v in losses.items():
                val_losses[k] += v
            count += 1
    for k in val_losses:
        val_losses[k] /= max(count, 1)
    out = dict([(k + "_valLoss", v) for k, v in val_losses.items()])
    out['lr'] = model.optimizers[0].param_groups[0]['lr']
    model.isTrain = bak_isTrain
    return out


def main(argv=None) -> dict:
    """validate.py:117-153: options -> dataset -> model -> validation -> averaged losses."""
    import time
    from .models import create_model
    from .options import parse
    opt = parse(argv)
    val_dataset = init_validation_dataloader(opt)
    print('Number of validation images = %d' % len(val_dataset))
    val_image_dir = join(opt.checkpoints_dir, opt.name, "val_visuals")
    model = create_model(opt)
    model.setup(opt)
    opt.isTrain = False
    model.isTrain = False
    t0 = time.time()
    val_losses = compute_validation(model, val_dataset, opt, val_image_dir=val_image_dir)
    dt = time.time() - t0
    print('c --------------------------------------- ')
    print('(validation, %d images, %.3f s, %.2f frames/s) ' % (len(val_dataset), dt, len(val_dataset) / max(dt, 1e-9))
          + ', '.join('%s: %.3f' % kv for kv in val_losses.items()))
    return val_losses


if __name__ == '__main__':
    main()
