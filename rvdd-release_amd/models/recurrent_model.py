"""``--model recurrent``: the reference's recurrentModel surface
(models/recurrent_model.py) with the per-frame hot loop in librvdd_hip.so.

Only the test branch of ``forward`` exists (TD=1, D=1, one unrolling per
call, models/recurrent_model.py:161-349); the recurrence state (previous
output, previous features) lives in device buffers owned by the runtime handle
and is reset by ``FirstOfVideo``.
"""
import torch

from .base_model import BaseModel
from .. import networks


class recurrentModel(BaseModel):
    @staticmethod
    def modify_commandline_options(parser, is_train=True):
        if is_train:
            parser.set_defaults(patch_depth=5, no_val=False, patch_width=136, val_dataset_mode='infer4rec')
        parser.add_argument('--model_patch_depth', type=int, default=2)
        parser.add_argument('--unroll_focus', type=str, default="gradual04_from20")
        parser.add_argument('--feature_rec', action='store_true', default=False)
        parser.add_argument('--prev_noisy_frame', action='store_true', default=False)
        parser.add_argument('--warp_raw', action='store_true', default=False)
        return parser

    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        for flag in ('no_predemosaic', 'raw_gt'):
            if getattr(opt, flag, False):
                raise NotImplementedError(f"rvdd: --{flag} is outside the built hot path "
                                          "(no BASELINE configuration uses it)")
        if opt.model_patch_depth != 2 or opt.input_nc != 3 or opt.output_nc != 3:
            raise NotImplementedError("rvdd: only --model_patch_depth 2, input_nc 3, output_nc 3 are built")
        # if only 1 unrolling was done in training the model is non-recurrent at
        # test time (recurrent_model.py:47-49, 233-235)
        self.training_unrollings = opt.patch_depth - opt.model_patch_depth + 1
        self.loss_names = ['L1', 'PSNR', 'Denoiser']
        self.visual_names = ['denoised']
        self.model_names = ['Denoise']
        network_input_nc = (opt.model_patch_depth + opt.future_patch_depth) * opt.input_nc
        self.netDenoise = networks.define_net_arch(
            network_input_nc, opt.output_nc, opt.netDenoiser, opt.init_type, opt.init_gain, self.gpu_ids,
            NoPF=opt.model_patch_depth - 1)
        self._netDenoise = self.netDenoise
        if bool(opt.feature_rec) != hasattr(self._netDenoise, 'NoPF'):
            raise ValueError("--feature_rec must be given exactly with the +feat / mode=feat networks")
        self.gt_nc = opt.input_nc
        self.data_nc = 4
        self._rt = None

    def to_device(self, x):
        return x.to(self.device, dtype=torch.float32, non_blocking=self.opt.non_blocking)

    def set_input(self, input):
        """recurrent_model.py:105-135.  The demosaic and flow upsample that the
        reference does here run inside ``rvdd_step`` (fused with the warp)."""
        self.n = self.to_device(input['n'])
        self.gt = self.to_device(input['gt'])
        self.image_paths = input['n_path']
        self.first_frame = False if self.isTrain else input['FirstOfVideo']
        # --no_warp: the dataset yields no flows and the reference never reads them (recurrent_model.py:117-122)
        self.flow = None if self.opt.no_warp else self.to_device(input['flow'])

    def forward(self):
        if self.isTrain:
            raise NotImplementedError("rvdd is an inference runtime; set model.isTrain = False "
                                      "(validate.py:137-138) before test()")
        B, C, h, w = self.n.shape
        fD = self.opt.future_patch_depth
        no_warp = bool(self.opt.no_warp)
        if C != 4 * (2 + fD) or (not no_warp and self.flow.shape[1] != 1 + fD):
            raise RuntimeError(f"input 'n' has {C} channels / 'flow' {None if no_warp else tuple(self.flow.shape)}; "
                               f"expected {4 * (2 + fD)} raw channels and {1 + fD} flows")
        rt = self._netDenoise.runtime_for(B, 2 * h, 2 * w, pin=True)
        if rt is not self._rt:
            self._rt = rt
            rt.set_option("no_warp", int(no_warp))
            rt.set_option("prev_noisy_frame", int(bool(self.opt.prev_noisy_frame)))
            rt.set_option("warp_raw", int(bool(self.opt.warp_raw)))
            rt.reset()
        if self.training_unrollings == 1 or self.first_frame:
            rt.reset()
        n, fl = self.n, self.flow
        self.denoised = rt.step(n[:, 0:4], n[:, 4:8], n[:, 8:12] if fD else None,
                                None if no_warp else fl[:, 0], fl[:, 1] if (fD and not no_warp) else None)

    def compute_losses(self):
        """Test branch of recurrent_model.py:512-525."""
        gt_2 = self.gt[:, -self.gt_nc:, :, :]
        l1, p = self._rt.psnr_l1(self.denoised, gt_2)
        self.loss_L1 = l1 * (self.opt.lambda_L1 / 100.0)
        self.loss_PSNR = p
        self.loss_Denoiser = self.loss_L1
