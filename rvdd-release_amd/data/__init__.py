"""data/__init__.py of the reference: `create_dataset(opt)` returns an iterable that yields batched
dicts like the reference's torch DataLoader wrapper (data/__init__.py:46-98) at the test-time settings
validate.py forces (batch_size 1, serial order, no workers: validate.py:46-48): tensors gain a leading
batch dimension, strings become 1-element lists."""
import importlib

import torch


def find_dataset_using_name(dataset_name):
    """data/__init__.py:18-37: data/<name>_dataset.py must hold a class <name>Dataset (case-insensitive)."""
    lib = importlib.import_module(f"{__name__}.{dataset_name}_dataset")
    target = dataset_name.replace('_', '') + 'dataset'
    for name, cls in lib.__dict__.items():
        if name.lower() == target.lower() and isinstance(cls, type):
            return cls
    raise NotImplementedError(f"In {dataset_name}_dataset.py, there should be a class {target} in lowercase.")


def _collate(sample):
    out = {}
    for k, v in sample.items():
        if torch.is_tensor(v):
            out[k] = v[None]
        elif isinstance(v, str):
            out[k] = [v]
        else:
            out[k] = v
    return out


class CustomDatasetDataLoader:
    def __init__(self, opt):
        if getattr(opt, 'batch_size', 1) != 1 or not getattr(opt, 'serial_batches', True):
            raise NotImplementedError("rvdd: the test-time loader is batch_size 1, serial order (validate.py:46-48)")
        self.opt = opt
        self.dataset = find_dataset_using_name(opt.dataset_mode)(opt)
        print("dataset [%s] was created" % type(self.dataset).__name__)

    def load_data(self):
        return self

    def __len__(self):
        return int(min(len(self.dataset), self.opt.max_dataset_size))

    def __iter__(self):
        for i in range(len(self.dataset)):
            if i >= self.opt.max_dataset_size:
                break
            yield _collate(self.dataset[i])

    def prepare_epoch(self):
        self.dataset.prepare_epoch()


def create_dataset(opt):
    return CustomDatasetDataLoader(opt).load_data()
