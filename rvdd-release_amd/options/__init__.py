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
        # dataset side (base_options.py:36-52, train_options.py:34-36, data/infer4rec_dataset.py:34-38)
        dataroot='./datasets/train_dataset', nFolder='noisy', gtFolder='gt', gt_linear_RGB_Folder='gt_linear_RGB',
        wFolder='warped', flowFolder='flow', raw_linear_RGB_Folder='raw_linear_RGB', check_data=True, videos=None,
        dataset_mode='axel4rec', serial_batches=False, num_threads=4, max_dataset_size=90000,
        val_dataroot='./datasets/validation_dataset', val_dataset_mode='infer4rec', val_videos='000,001,002,003,004',
        crop_data=None, warpeddata=False,
    )
    for k, v in overrides.items():
        if not hasattr(opt, k):
            raise AttributeError(f"unknown option {k!r}")
        setattr(opt, k, v)
    _finish(opt)
    return opt


def _finish(opt):
    warpstr = '-warp' if not opt.no_warp else ''
    suffixstr = "-" + opt.suffix if opt.suffix else ""
    opt.name = "%s-%s%s-i%do%d%s" % (opt.model, opt.netDenoiser, warpstr, opt.input_nc, opt.output_nc,
                                     suffixstr)          # base_options.py:131-136


def parse(argv=None) -> Namespace:
    """The reference's command line for the inference path (`validate.py` parses TrainOptions): every
    option of `make_opt` under the reference's flag name; switches are store_true like there."""
    import argparse
    defaults = vars(make_opt())
    parser = argparse.ArgumentParser(description="rvdd validate (reference: validate.py)")
    for k, v in defaults.items():
        if k in ('name', 'isTrain'):
            continue
        if k == 'gpu_ids':
            parser.add_argument('--gpu_ids', type=str, default='0')
        elif isinstance(v, bool):
            parser.add_argument('--' + k, action='store_true', default=v)
        elif v is None:
            parser.add_argument('--' + k, type=str, default=None)
        else:
            parser.add_argument('--' + k, type=type(v), default=v)
    ns = parser.parse_args(argv)
    ns.gpu_ids = [int(i) for i in ns.gpu_ids.split(',') if int(i) >= 0]      # base_options.py:139-146
    ns.isTrain = True
    _finish(ns)
    return ns
