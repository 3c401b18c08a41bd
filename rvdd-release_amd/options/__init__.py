"""The hot-path subset of the reference's option namespace.

The reference builds ``opt`` with argparse in three stages
(options/base_options.py:74-100); the inference path reads only the
attributes below.  ``make_opt`` returns a namespace with the reference's
defaults (base_options.py:24-68, train_options.py:13-38,
recurrent_model.py:27-36) so that ``create_model(opt)`` / ``model.setup(opt)``
can be driven exactly like validate.py:117-138 does.
"""
from argparse import Namespace


def make_opt(**overrides) -> Namespace:
    opt = Namespace(
        gpu_ids=[0], checkpoints_dir='./checkpoints', model='recurrent', input_nc=3, output_nc=3,
        netDenoiser='convunet-mode=fixedfeatures', init_type='kaiming', init_gain=0.02, bit_depth=12,
        no_warp=False, warp_method='tvl1', non_blocking=True, batch_size=1,
        patch_depth=5,                       # recurrent_model.py:28 set_defaults(patch_depth=5)
        future_patch_depth=0, epoch='latest_val', verbose=False, suffix='', no_predemosaic=False,
        raw_gt=False, val_flow_from_denoised=False, model_patch_depth=2, feature_rec=False,
        prev_noisy_frame=False, warp_raw=False, path2epoch='', lambda_L1=100.0, lr=0.00016,
        isTrain=True,                        # validate.py parses TrainOptions (isTrain=True at parse time)
    )
    for k, v in overrides.items():
        if not hasattr(opt, k):
            raise AttributeError(f"unknown option {k!r}")
        setattr(opt, k, v)
    warpstr = '-warp' if not opt.no_warp else ''
    suffixstr = "-" + opt.suffix if opt.suffix else ""
    opt.name = "%s-%s%s-i%do%d%s" % (opt.model, opt.netDenoiser, warpstr, opt.input_nc, opt.output_nc,
                                     suffixstr)          # base_options.py:131-136
    return opt
