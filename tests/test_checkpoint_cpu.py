"""nmf_amd/checkpoint.py: checkpoints written by the REFERENCE pickle an OmegaConf DictConfig (modules/tensor_nerf.py:120-134)
and must load without omegaconf and without running pickled code.  omegaconf is absent offline, so the
fixture is produced by a structural mimic of its classes (same module / class names and the `__getstate__` layout of
omegaconf 2.x: containers keep `_content` = dict / list of nodes, value nodes keep `_val`, all carry `_metadata` objects and
a `_parent` back-reference); the mimic is removed from sys.modules before the file is read."""
import pickle
import sys
import types
import typing

import pytest
import torch


def _install_mimic():
    mods = {n: types.ModuleType(n) for n in ("omegaconf", "omegaconf.base", "omegaconf.nodes", "omegaconf.dictconfig",
                                             "omegaconf.listconfig")}

    def cls(mod, name, base=object):
        c = type(name, (base,), {"__module__": mod})
        setattr(mods[mod], name, c)
        return c

    Metadata = cls("omegaconf.base", "Metadata")
    ContainerMetadata = cls("omegaconf.base", "ContainerMetadata", Metadata)
    nodes = {t: cls("omegaconf.nodes", n) for t, n in ((str, "StringNode"), (int, "IntegerNode"), (float, "FloatNode"),
                                                        (bool, "BooleanNode"), (type(None), "AnyNode"))}
    DictConfig = cls("omegaconf.dictconfig", "DictConfig")
    ListConfig = cls("omegaconf.listconfig", "ListConfig")

    def meta(kind, key, ref=typing.Any):
        m = kind()
        m.__dict__.update(ref_type=ref, object_type=None, optional=True, key=key, flags=None, flags_root=False,
                          resolver_cache={})
        if kind is ContainerMetadata:
            m.__dict__.update(key_type=typing.Any, element_type=typing.Any)
        return m

    def wrap(v, parent, key):
        if isinstance(v, dict):
            n = DictConfig()
            n.__dict__.update(_metadata=meta(ContainerMetadata, key), _parent=parent, _flags_cache=None)
            n.__dict__["_content"] = {k: wrap(x, n, k) for k, x in v.items()}
            n.__dict__["_metadata"].object_type = dict
            return n
        if isinstance(v, (list, tuple)):
            n = ListConfig()
            n.__dict__.update(_metadata=meta(ContainerMetadata, key), _parent=parent, _flags_cache=None)
            n.__dict__["_content"] = [wrap(x, n, i) for i, x in enumerate(v)]
            return n
        n = nodes.get(type(v), nodes[type(None)])()
        n.__dict__.update(_metadata=meta(Metadata, key), _parent=parent, _val=v)
        return n

    sys.modules.update(mods)
    return lambda v: wrap(v, None, None), list(mods)


def _remove(names):
    for n in names:
        sys.modules.pop(n, None)


ARCH = {"_target_": "modules.tensor_nerf.TensorNeRF", "_partial_": True, "eval_batch_size": 4096, "hdr": False,
        "recur_stepmul": 0.5, "rf": {"_target_": "fields.tensoRF.TensorVMSplit", "_partial_": True, "grid_size": [128, 128, 128],
                                     "lr": 0.02, "activation": "softplus"},
        "model": {"brdf": {"bias": 0.123, "h_encoder": {"_target_": "modules.ish.ListISH", "degs": [0, 1, 2, 4]}},
                  "diffuse_module": {"diffuse_bias": -0.456, "roughness_bias": -1.25}, "max_retrace_rays": [1000],
                  "visibility_module": None}}


def test_reference_style_checkpoint_loads_without_omegaconf(tmp_path):
    from nmf_amd.checkpoint import load_checkpoint
    wrap, names = _install_mimic()
    sd = {"rf.aabb": torch.tensor([[-1.5] * 3, [1.5] * 3]), "rf.grid_size": torch.tensor([128, 128, 128]),
          "rf.density_rf.app_plane.0": torch.randn(1, 16, 8, 8), "bg_module.mipbias": torch.tensor(1.0, dtype=torch.float64)}
    path = str(tmp_path / "ref.th")
    torch.save({"config": wrap(ARCH), "state_dict": sd}, path)
    _remove(names)
    assert "omegaconf" not in sys.modules
    with pytest.raises(Exception):
        torch.load(path, weights_only=True)                       # the plain safe loader refuses the omegaconf globals
    ck = load_checkpoint(path)
    assert ck["config"] == ARCH and type(ck["config"]) is dict and type(ck["config"]["rf"]["grid_size"]) is list
    assert set(ck["state_dict"]) == set(sd) and all(torch.equal(ck["state_dict"][k], v) for k, v in sd.items())
    assert ck["state_dict"]["bg_module.mipbias"].dtype == torch.float64
    assert "omegaconf" not in sys.modules


def test_own_checkpoints_take_the_weights_only_path_and_code_is_refused(tmp_path):
    from nmf_amd.checkpoint import load_checkpoint
    path = str(tmp_path / "own.th")
    torch.save({"config": ARCH, "state_dict": {"a": torch.arange(3)}}, path)
    ck = load_checkpoint(path)
    assert ck["config"] == ARCH and torch.equal(ck["state_dict"]["a"], torch.arange(3))

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > /dev/null",))

    bad = str(tmp_path / "bad.th")
    torch.save({"config": Evil(), "state_dict": {}}, bad)
    with pytest.raises(pickle.UnpicklingError, match="refused"):
        load_checkpoint(bad)


_GADGET_RAN = []


def _gadget_payload(*a):
    _GADGET_RAN.append(a)
    return {}


class _InnerEvil:
    def __reduce__(self):
        return (_gadget_payload, ("inner",))


def test_load_from_bytes_gadget_is_refused(tmp_path):
    """`torch.storage._load_from_bytes(b)` is `torch.load(BytesIO(b), weights_only=False)` -- a checkpoint that
    names it smuggles a second, unrestricted pickle through the restricted reader.  It is not on the allowlist any more;
    nor is anything matched by prefix / suffix (`_rebuild*`, `*Storage`)."""
    import io
    from nmf_amd.checkpoint import load_checkpoint
    inner = io.BytesIO()
    torch.save(_InnerEvil(), inner)

    class Outer:
        def __reduce__(self):
            return (torch.storage._load_from_bytes, (inner.getvalue(),))

    class Wide:                       # a `_rebuild*` name that is not one of the four tensor constructors
        def __reduce__(self):
            return (torch._utils._rebuild_wrapper_subclass, ())

    for i, obj in enumerate((Outer(), Wide())):
        bad = str(tmp_path / f"gadget{i}.th")
        torch.save({"config": obj, "state_dict": {"a": torch.arange(3)}}, bad, pickle_protocol=4)
        with pytest.raises(pickle.UnpicklingError, match="refused"):
            load_checkpoint(bad)
    assert not _GADGET_RAN
