"""The validation driver of the reference (validate.py) on the HIP runtime.

* `compute_validation(model, val_dataset, opt, ...)` -- validate.py:54-114: per output frame `set_input` /
  `test` / `compute_losses`, `FirstOfVideo` latched on a change of video folder, losses averaged over the
  dataset and returned as {'L1_valLoss', 'PSNR_valLoss', 'Denoiser_valLoss', 'lr'}.
* `compute_flows_from_denoised(data, model, opt)` -- validate.py:16-38 (`--val_flow_from_denoised`): the flow
  towards the previous DENOISED frame, recomputed online: remosaick -> TV-L1 on the device (`rvdd_tvl1flow`).
* `init_validation_dataloader(opt)` -- validate.py:40-52; `main(argv)` -- validate.py:117-153, so that
  `python -m rvdd_release_amd.validate <the flags of scripts/test-*.sh>` reads frames and flows from disk and
  writes `<checkpoints_dir>/<name>/val_visuals/<video>/<frame>_denoised.tif` + `output.log`.

`val_dataset` is any iterable of the dicts the reference's `infer4recDataset` yields through its loader
(data/infer4rec_dataset.py:226-230): 'n' [1,(2+f)*4,h,w], 'gt' [1,6,H,W], 'flow' [1,1+f,2,h,w], 'n_path', 'gt_path'.
"""
from __future__ import annotations

import copy
import os
import time
from typing import Callable, Dict, Iterable, Optional

import torch

from .util._ops import ops_runtime
from .util.Hamilton_Adam_demo import HamiltonAdam


def compute_flows_from_denoised(data: dict, model, opt) -> None:
    """Replace the flow towards the previous frame in data['flow'] by the TV-L1 flow from the current noisy frame
    to the re-mosaicked previous OUTPUT.

    The released reference cannot actually run this branch (it hands `remosaick` a squeezed 3-D tensor,
    validate.py:29-31 vs util/Hamilton_Adam_demo.py:237-238, and appends the same flow `opt.patch_depth - 1`
    times); this is its evident intent: ONE flow, shaped like the dataset's `data['flow']` ([1,1,2,h,w]).
    With a future frame (`--future_patch_depth 1`) the reference would in addition pair the NEXT noisy frame
    (`data['n'][0, -4:]`) with the previous output and leave the model without its second flow (an IndexError in
    recurrent_model.py:317); here the current frame is the second packed frame whatever follows it, and the flow
    towards the next frame -- which no output exists for yet -- stays the dataset's pre-computed one."""
    dev = model.device
    noisy_cur = data['n'][0, 4:8, :, :].to(dev, torch.float32)                       # packed raw, [-1,1]
    prev_out = HamiltonAdam('gbrg').remosaick(model.denoised.to(dev))[0]
    # the reference hands (x+1)/2 images to CPPbridge, which reduces 4 channels to their mean (library.py:67-68, :165-167)
    target = ((noisy_cur + 1.0) / 2.0).mean(dim=0).contiguous()
    moving = ((prev_out + 1.0) / 2.0).mean(dim=0).contiguous()
    flow = ops_runtime(dev.index or 0).tvl1flow(target, moving)                     # moving(x + flow) ~ target(x)
    if getattr(opt, "future_patch_depth", 0):
        keep = data['flow'][:, 1:2].to(dev, torch.float32)                          # cur -> next, from the dataset
        data['flow'] = torch.cat((flow[None, None], keep), dim=1)
    else:
        data['flow'] = flow[None, None]


def init_validation_dataloader(opt):
    """validate.py:40-52: the validation view of the options, then the dataset."""
    from .data import create_dataset
    v = copy.deepcopy(opt)
    v.dataroot, v.dataset_mode, v.videos = opt.val_dataroot, opt.val_dataset_mode, opt.val_videos
    v.max_dataset_size = float("inf")
    v.num_threads, v.batch_size, v.serial_batches = 0, 1, True
    if hasattr(opt, 'model_patch_depth'):
        v.patch_depth = opt.model_patch_depth
    return create_dataset(v)


def compute_validation(model, val_dataset: Iterable[Dict], opt, on_frame: Optional[Callable] = None,
                       val_image_dir: Optional[str] = None) -> dict:
    """validate.py:54-114.  With `val_image_dir` (and a dataset made by `create_dataset`) every frame is written
    to <val_image_dir>/<video>/<frame>_denoised.tif and its losses appended to output.log, as the reference does;
    `on_frame(i, data, visuals, losses)` is an in-memory hook beside that."""
    online_flow = (not model.isTrain) and bool(getattr(opt, "val_flow_from_denoised", False)) and not opt.no_warp
    was_training = model.isTrain
    model.isTrain = False
    model.eval()
    totals = {name: 0.0 for name in model.get_current_losses()}
    frames = 0
    previous_video = ''
    with torch.no_grad():
        for i, data in enumerate(val_dataset):
            video = os.path.dirname(data['gt_path'][0])
            data['FirstOfVideo'] = video != previous_video
            previous_video = video
            if online_flow and not data['FirstOfVideo']:
                compute_flows_from_denoised(data, model, opt)
            model.set_input(data)
            model.test()
            model.compute_losses()
            losses = model.get_current_losses()
            if on_frame is not None:
                on_frame(i, data, model.get_current_visuals(), losses)
            if val_image_dir is not None:
                _write_frame(model, val_dataset, val_image_dir, i, losses)
            for name, value in losses.items():
                totals[name] += value
            frames += 1
    result = {name + "_valLoss": total / max(frames, 1) for name, total in totals.items()}
    result['lr'] = model.optimizers[0].param_groups[0]['lr']
    model.isTrain = was_training
    return result


def _write_frame(model, val_dataset, val_image_dir, i, losses):
    """validate.py:88-103: the visuals as TIFF under the video's folder, the losses as one line of output.log."""
    from .library import pathdiff, print_dict
    from .util.visualizer import save_images
    img_path = model.get_image_paths()
    if i % 40 == 0:
        print('processing (%04d)-th image... %s' % (i, img_path))
    save_images(val_image_dir, model.get_current_visuals(), [os.path.basename(img_path[0])],
                subfolder=pathdiff(img_path[0], val_dataset.dataset.n_paths))
    print_dict(losses, suffix="", savefile=os.path.join(val_image_dir, "output.log"))


def main(argv=None) -> dict:
    """validate.py:117-153: options -> dataset -> model -> validation -> averaged losses."""
    from .models import create_model
    from .options import parse
    opt = parse(argv)
    val_dataset = init_validation_dataloader(opt)
    print('Number of validation images = %d' % len(val_dataset))
    model = create_model(opt)
    model.setup(opt)
    opt.isTrain = model.isTrain = False
    t0 = time.time()
    val_losses = compute_validation(model, val_dataset, opt,
                                    val_image_dir=os.path.join(opt.checkpoints_dir, opt.name, "val_visuals"))
    dt = time.time() - t0
    print('(validation, %d images, %.3f s, %.2f frames/s) ' % (len(val_dataset), dt, len(val_dataset) / max(dt, 1e-9))
          + ', '.join('%s: %.3f' % kv for kv in val_losses.items()))
    return val_losses


if __name__ == '__main__':
    main()
