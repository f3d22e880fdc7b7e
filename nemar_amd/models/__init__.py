"""Model discovery by name — mirror of reference models/__init__.py:25-67: `--model X` resolves to the module
`<this package>.X_model` and, inside it, the BaseModel subclass whose lower-cased name is `Xmodel`."""
import importlib

from .base_model import BaseModel


def find_model_using_name(model_name):
    model_filename = __name__ + "." + model_name + "_model"
    modellib = importlib.import_module(model_filename)
    model = None
    target_model_name = model_name.replace('_', '') + 'model'
    for name, cls in modellib.__dict__.items():
        if name.lower() == target_model_name.lower() and isinstance(cls, type) and issubclass(cls, BaseModel):
            model = cls
    if model is None:
        print("In %s.py, there should be a subclass of BaseModel with class name that matches %s in lowercase." % (
            model_filename, target_model_name))
        exit(0)
    return model


def get_option_setter(model_name):
    return find_model_using_name(model_name).modify_commandline_options


def create_model(opt):
    model = find_model_using_name(opt.model)
    instance = model(opt)
    print("model [%s] was created" % type(instance).__name__)
    return instance
