"""``--model`` registry (models/__init__.py:25-67 of the reference)."""
import importlib

from .base_model import BaseModel


def find_model_using_name(model_name):
    modellib = importlib.import_module(f"{__name__}.{model_name}_model")
    target = model_name.replace('_', '') + 'model'
    model = None
    for name, cls in modellib.__dict__.items():
        if name.lower() == target.lower() and isinstance(cls, type) and issubclass(cls, BaseModel):
            model = cls
    if model is None:
        raise NotImplementedError(f"In {model_name}_model.py, there should be a subclass of BaseModel "
                                  f"with class name that matches {target} in lowercase.")
    return model


def create_model(opt):
    model = find_model_using_name(opt.model)
    instance = model(opt)
    print("model [%s] was created" % type(instance).__name__)
    return instance
