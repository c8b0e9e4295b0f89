"""Drop-in glue for files and scripts written against the reference package.

* `load_checkpoint(path)`: the reference saves `{"config": EasyDict, "model": state_dict, "optimizer_*",
  "scheduler_*", "iteration", "avg_val_loss"}` with torch.save (scripts/train.py:219-231) and reads it back with a
  plain torch.load (scripts/test.py:78), which unpickles `easydict.EasyDict` objects.  `easydict` is not a
  dependency of this package, so the checkpoint is read through a restricted unpickler that maps
  `easydict.EasyDict` to `agdiff_amd.Config` (same attribute / item access) and otherwise admits only tensors and
  plain containers.
* `install()`: registers `agdiff.models.epsnet` (the module scripts/test.py:21 imports `get_model` from) in
  sys.modules as an alias of this package's model module, so that the reference's driver scripts run unedited.
"""
import pickle
import sys
import types

from .config import Config

# Exact (module, name) pairs a reference checkpoint pickles (scripts/train.py:219-231: a state_dict, two optimizer and two
# scheduler state_dicts, an int and a float, the EasyDict config).  Nothing is admitted by module prefix: a pickle may name
# ANY attribute reachable from an admitted module ("torch.serialization" -> os), so every global is listed by hand.
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
    ("_codecs", "encode"),
}
_SAFE_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "complex", "bool", "str", "bytes",
                  "bytearray", "slice", "range"}
_TORCH_STORAGES = {"DoubleStorage", "FloatStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage",
                   "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage", "UntypedStorage"}


class _Unpickler(pickle.Unpickler):
    """Allow-list unpickler: torch tensor rebuild helpers, torch storages / dtypes, numpy scalars / arrays, std
    containers, and easydict.EasyDict -> agdiff_amd.Config, each by its exact (module, name).  Anything else -- dotted
    names, getattr, other attributes of the admitted modules -- is refused (pass trust=True to load_checkpoint for a
    file you trust that pickles other classes)."""

    def find_class(self, module, name):
        if (module, name) in (("easydict", "EasyDict"), ("agdiff_amd.config", "Config")):
            return Config
        ok = False
        if "." not in name:
            if module == "builtins":
                ok = name in _SAFE_BUILTINS
            elif (module, name) in _SAFE_GLOBALS:
                ok = True
            elif module == "torch":
                import torch
                ok = name in _TORCH_STORAGES or isinstance(getattr(torch, name, None), torch.dtype)
        if ok:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("checkpoint pickles %s.%s, which agdiff_amd.compat.load_checkpoint does not admit; "
                                     "pass trust=True if the file comes from a source you trust" % (module, name))


class _TrustingUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "easydict" and name == "EasyDict":
            return Config
        return super().find_class(module, name)


def _pickle_module(unpickler):
    m = types.ModuleType("agdiff_amd._ckpt_pickle")
    m.Unpickler = unpickler
    m.load = lambda f, **kw: unpickler(f, **kw).load()
    m.loads = pickle.loads
    m.__name__ = "pickle"
    return m


def load_checkpoint(path, map_location="cpu", trust=False):
    """torch.load for reference checkpoints (scripts/train.py:219-231) without the `easydict` package."""
    import torch
    return torch.load(path, map_location=map_location, weights_only=False,
                      pickle_module=_pickle_module(_TrustingUnpickler if trust else _Unpickler))


def model_config(ckpt):
    """`ckpt["config"].model` (scripts/test.py:111) for EasyDict-derived, Config and plain-dict checkpoints."""
    cfg = ckpt["config"]
    if isinstance(cfg, dict) and not isinstance(cfg, Config):
        cfg = Config(cfg)
    return cfg.model


def install(force=True):
    """Make `from agdiff.models.epsnet import get_model` (scripts/test.py:21, scripts/train.py) resolve to this
    package.  If the reference package is importable its other sub-modules (utils.datasets, utils.transforms, ...)
    stay as they are; only `agdiff.models.epsnet` is replaced.  Also provides `easydict.EasyDict` (= Config) when
    the easydict package is absent, so that a plain torch.load(ckpt) of a reference checkpoint works too."""
    from . import epsnet
    shim = types.ModuleType("agdiff.models.epsnet")
    shim.__doc__ = "alias of agdiff_amd.epsnet installed by agdiff_amd.compat.install()"
    for name in ("get_model", "DualEncoderEpsNetwork", "get_beta_schedule", "is_local_edge", "is_radius_edge"):
        setattr(shim, name, getattr(epsnet, name))
    shim.__all__ = ["get_model", "DualEncoderEpsNetwork"]
    dualenc = types.ModuleType("agdiff.models.epsnet.dualenc")
    for name in ("DualEncoderEpsNetwork", "get_beta_schedule", "is_local_edge", "is_radius_edge"):
        setattr(dualenc, name, getattr(epsnet, name))
    shim.dualenc = dualenc
    for pkg in ("agdiff", "agdiff.models"):
        if pkg not in sys.modules:
            try:
                __import__(pkg)
            except Exception:
                mod = types.ModuleType(pkg)
                mod.__path__ = []
                sys.modules[pkg] = mod
    if force or "agdiff.models.epsnet" not in sys.modules:
        sys.modules["agdiff.models.epsnet"] = shim
        sys.modules["agdiff.models.epsnet.dualenc"] = dualenc
        sys.modules["agdiff.models"].epsnet = shim
    if "agdiff" in sys.modules and not hasattr(sys.modules["agdiff"], "models"):
        sys.modules["agdiff"].models = sys.modules["agdiff.models"]
    try:
        import easydict  # noqa: F401
    except Exception:
        ed = types.ModuleType("easydict")
        ed.EasyDict = Config
        sys.modules["easydict"] = ed
    return shim
