"""The part of models/base_model.py (reference) that validate.py exercises."""
import copy
import os
from abc import ABC, abstractmethod
from collections import OrderedDict

import torch


class _NoOptimizer:
    """validate.py:109 reads ``model.optimizers[0].param_groups[0]['lr']``."""

    def __init__(self, lr):
        self.param_groups = [{'lr': lr}]


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

    def setup(self, opt):
        """base_model.py:87-99: load the networks named by the options."""
        if not self.isTrain:
            self.load_networks(opt.epoch)
        if self.isTrain and opt.path2epoch != '':
            self.load_networks(opt.path2epoch, pathepoch=True)
        self.print_networks(getattr(opt, 'verbose', False))

    def train(self):
        raise NotImplementedError("rvdd is an inference runtime; training is out of scope")

    def eval(self):
        for name in self.model_names:
            getattr(self, 'net' + name).eval()

    @torch.no_grad()
    def test(self):
        self.forward()

    def get_image_paths(self):
        return self.image_paths

    def get_current_visuals(self):
        visual_ret = OrderedDict()
        for name in self.visual_names:
            visual_ret[name] = getattr(self, name)
        return visual_ret

    def get_current_losses(self):
        errors_ret = OrderedDict()
        for name in self.loss_names:
            loss_str = 'loss_' + name
            errors_ret[name] = float(getattr(self, loss_str)) if hasattr(self, loss_str) else 0
        return errors_ret

    def load_networks(self, epoch, pathepoch=False):
        """base_model.py:173-196.  Reads ``<epoch>_net_<name>.pth`` (a torch
        state_dict) or, when that file is absent, its ``.safetensors`` twin
        ``<epoch>.safetensors`` (the neutral conversion shipped in weights/)."""
        for name in self.model_names:
            if not pathepoch:
                stem = os.path.join(self.save_dir, '%s' % epoch)
            else:
                stem = '%s' % epoch
            load_path = '%s_net_%s.pth' % (stem, name)
            net = getattr(self, 'net' + name)
            if os.path.exists(load_path):
                print('loading the model from %s' % load_path)
                state_dict = torch.load(load_path, map_location='cpu', weights_only=True)
            elif os.path.exists(stem + '.safetensors'):
                from safetensors.torch import load_file
                print('loading the model from %s' % (stem + '.safetensors'))
                state_dict = load_file(stem + '.safetensors')
            else:
                raise FileNotFoundError(load_path)
            if hasattr(state_dict, '_metadata'):
                del state_dict._metadata
            net.load_state_dict(state_dict)

    def print_networks(self, verbose):
        print('---------- Networks initialized -------------')
        for name in self.model_names:
            net = getattr(self, 'net' + name)
            num_params = sum(p.numel() for p in net.parameters())
            print('[Network %s] Total number of parameters : %.3f M' % (name, num_params / 1e6))
        print('-----------------------------------------------')
