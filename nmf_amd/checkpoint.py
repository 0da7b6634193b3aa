"""Reading checkpoints -- the build's own and the REFERENCE's (modules/tensor_nerf.py:120-134: `torch.save({"config":
args.model.arch, "state_dict": ...})`, where the config is an OmegaConf DictConfig object, i.e. the file pickles
`omegaconf.*` classes that are not installed here and should not be needed to read a dictionary of numbers).

load_checkpoint(path):
  1. `torch.load(..., weights_only=True)`: plain containers + tensors (everything this package writes).
  2. otherwise a RESTRICTED unpickler: tensors / storages, builtin containers, `typing.Any`, and -- for every global of the
     `omegaconf` package -- an inert stand-in that only records the pickled state (no imports, no calls into foreign code;
     any other global is refused).  The stand-ins are then folded into plain dict / list / scalars:
         DictConfig.__dict__['_content']  = {key: node}      ListConfig.__dict__['_content'] = [node, ...]
         ValueNode.__dict__['_val']       = python value     (omegaconf/basecontainer.py `__getstate__`, omegaconf/nodes.py)
"""
import collections
import pickle
import types
import typing

import torch


class _Inert:
    """stand-in for an omegaconf class: keeps whatever state the pickle hands over"""

    def __init__(self, *a, **k):
        self.__dict__["_args"] = a

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):      # (dict state, slots state)
            state = dict(state[0] or {}, **state[1])
        self.__dict__.update(state if isinstance(state, dict) else {"_state": state})


_STANDINS = {}


def _standin(module, name):
    key = (module, name)
    if key not in _STANDINS:
        _STANDINS[key] = type(name, (_Inert,), {"__module__": module, "_oc_name": name})
    return _STANDINS[key]


_ALLOWED = {
    ("collections", "OrderedDict"): collections.OrderedDict, ("collections", "defaultdict"): collections.defaultdict,
    ("typing", "Any"): typing.Any, ("builtins", "dict"): dict, ("builtins", "list"): list, ("builtins", "tuple"): tuple,
    ("builtins", "set"): set, ("builtins", "frozenset"): frozenset, ("builtins", "int"): int, ("builtins", "float"): float,
    ("builtins", "str"): str, ("builtins", "bool"): bool, ("builtins", "bytes"): bytes, ("builtins", "complex"): complex,
    ("builtins", "object"): object, ("builtins", "slice"): slice, ("copyreg", "_reconstructor"): None,
}


# Exact names only.  `torch.storage._load_from_bytes` is deliberately absent: it is `torch.load(BytesIO(b), weights_only=False)`,
# i.e. a second, UNRESTRICTED unpickler reachable through one REDUCE; state dicts written by torch.save do not
# need it.  No prefix / suffix matching either: every callable an attacker could name has to be in this list.
_STORAGES = ("FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage", "ShortStorage",
             "CharStorage", "ByteStorage", "BoolStorage", "UntypedStorage")
_DTYPES = ("float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool")
_TENSOR_GLOBALS = (
    {("torch._utils", n) for n in ("_rebuild_tensor_v2", "_rebuild_tensor", "_rebuild_parameter",
                                   "_rebuild_parameter_with_state")}
    | {("torch", n) for n in _STORAGES + _DTYPES + ("Size", "device", "Tensor")}
    | {("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage")}
    | {(m, n) for m in ("numpy.core.multiarray", "numpy._core.multiarray") for n in ("_reconstruct", "scalar")}
    | {("numpy", "ndarray"), ("numpy", "dtype")})


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "omegaconf" or module.startswith("omegaconf."):
            return _standin(module, name)
        if module == "__builtin__":                       # protocol-2 spelling of builtins (torch.save's default protocol)
            module = "builtins"
        if (module, name) in _ALLOWED:
            if (module, name) == ("copyreg", "_reconstructor"):
                import copyreg
                return copyreg._reconstructor
            return _ALLOWED[(module, name)]
        if (module, name) in _TENSOR_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"checkpoint refers to {module}.{name}: refused (only tensors, plain containers and "
                                     "omegaconf configuration nodes are read)")


_pickle_module = types.SimpleNamespace(Unpickler=_RestrictedUnpickler, load=lambda f, **k: _RestrictedUnpickler(f, **k).load(),
                                       __name__="nmf_amd.checkpoint._pickle_module")


def to_plain(x):
    """omegaconf stand-ins (and containers holding them) -> dict / list / python scalars"""
    if isinstance(x, _Inert):
        d = x.__dict__
        name = type(x)._oc_name
        if "_content" in d:
            c = d["_content"]
            if isinstance(c, dict):
                return {to_plain(k): to_plain(v) for k, v in c.items()}
            if isinstance(c, (list, tuple)):
                return [to_plain(v) for v in c]
            return to_plain(c)                       # None / missing / interpolation string
        if "_val" in d:
            return to_plain(d["_val"])
        raise pickle.UnpicklingError(f"omegaconf object {name} without _content / _val: unknown pickle layout")
    if isinstance(x, dict):
        return {to_plain(k): to_plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)) and not isinstance(x, torch.Size):
        return type(x)(to_plain(v) for v in x) if isinstance(x, tuple) else [to_plain(v) for v in x]
    return x


def load_checkpoint(path, map_location="cpu"):
    """-> {"config": plain nested dict, "state_dict": {name: tensor}} for checkpoints of this package and of the reference"""
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except Exception as first:                                   # noqa: BLE001  (UnpicklingError wording varies)
        try:
            ck = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_pickle_module)
        except pickle.UnpicklingError:
            raise
        except Exception as second:                              # noqa: BLE001
            raise pickle.UnpicklingError(f"cannot read {path}: {first}; restricted reader: {second}") from second
    if isinstance(ck, dict) and "config" in ck:
        ck = dict(ck)
        ck["config"] = to_plain(ck["config"])
    return ck
