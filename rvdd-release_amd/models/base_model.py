"""Host-side base class of the model plugins: the slice of the reference's
``BaseModel`` interface (models/base_model.py:9-67, 87-153, 173-214) that
``validate.py`` and ``recurrentModel`` talk to.  Inference only: there are no
optimizers, schedulers or ``save_networks`` behind it."""
import copy
import os
from abc import ABC, abstractmethod
from collections import OrderedDict

import torch


class _NoOptimizer:
    """validate.py:109 reads ``model.optimizers[0].param_groups[0]['lr']``."""

    def __init__(self, lr):
        self.param_groups = [{'lr': lr}]


def _read_state_dict(stem, name):
    """``<stem>_net_<name>.pth`` (a torch state_dict, tensors only) or, when that file is absent, the neutral
    conversion ``<stem>.safetensors`` shipped in weights/."""
    pth = f"{stem}_net_{name}.pth"
    twin = stem + ".safetensors"
    if os.path.exists(pth):
        print('loading the model from %s' % pth)
        sd = torch.load(pth, map_location='cpu', weights_only=True)
        return OrderedDict((k, v) for k, v in sd.items())          # drops a pickled _metadata attribute, if any
    if os.path.exists(twin):
        from safetensors.torch import load_file
        print('loading the model from %s' % twin)
        return load_file(twin)
    raise FileNotFoundError(pth)


class BaseModel(ABC):
    def __init__(self, opt):
        self.opt = copy.deepcopy(opt)
        self.gpu_ids = opt.gpu_ids
        self.isTrain = opt.isTrain
        if not self.gpu_ids:
            raise RuntimeError("rvdd: --gpu_ids -1 (CPU) is not available; this runtime has no CPU path")
        self.device = torch.device(f'cuda:{self.gpu_ids[0]}')
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.loss_names = []
        self.model_names = []
        self.visual_names = []
        self.optimizers = [_NoOptimizer(getattr(opt, 'lr', 0.0))]
        self.image_paths = []
        self.best_val_score = float("inf")

    # -- what a model plugin supplies ----------------------------------------
    @abstractmethod
    def set_input(self, input):
        pass

    @abstractmethod
    def forward(self):
        pass

    @abstractmethod
    def compute_losses(self):
        pass

    def optimize_parameters(self):
        raise NotImplementedError("rvdd is an inference runtime; training is out of scope")

    def train(self):
        raise NotImplementedError("rvdd is an inference runtime; training is out of scope")

    # -- what validate.py calls -------------------------------------------------
    def _nets(self):
        return [(n, getattr(self, 'net' + n)) for n in self.model_names]

    def setup(self, opt):
        """Weights come from ``--epoch`` under the run directory at test time, from ``--path2epoch`` (a path stem)
        when the options were parsed in training mode, which is how validate.py parses them (base_model.py:87-99)."""
        if self.isTrain:
            if opt.path2epoch != '':
                self.load_networks(opt.path2epoch, pathepoch=True)
        else:
            self.load_networks(opt.epoch)
        self.print_networks(getattr(opt, 'verbose', False))

    def eval(self):
        for _, net in self._nets():
            net.eval()

    @torch.no_grad()
    def test(self):
        self.forward()

    def get_image_paths(self):
        return self.image_paths

    def get_current_visuals(self):
        """name -> tensor for every entry of ``visual_names`` (validate.py:88 saves 'denoised')."""
        return OrderedDict((name, getattr(self, name)) for name in self.visual_names)

    def get_current_losses(self):
        """name -> Python float of ``self.loss_<name>``; 0 until the first compute_losses (validate.py:69,101)."""
        return OrderedDict((name, float(getattr(self, 'loss_' + name, 0))) for name in self.loss_names)

    def load_networks(self, epoch, pathepoch=False):
        """base_model.py:173-196, but strict: the runtime rejects unknown and missing keys where the reference
        loads with strict=False."""
        stem = str(epoch) if pathepoch else os.path.join(self.save_dir, str(epoch))
        for name, net in self._nets():
            net.load_state_dict(_read_state_dict(stem, name))

    def print_networks(self, verbose):
        print('---------- Networks initialized -------------')
        for name, net in self._nets():
            count = sum(p.numel() for p in net.parameters())
            print('[Network %s] Total number of parameters : %.3f M' % (name, count / 1e6))
        print('-----------------------------------------------')
