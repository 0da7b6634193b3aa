"""Config surface without hydra -- counterpart of the reference's `@hydra.main(config_path="configs",
config_name="default")` entry (train.py:904-911) for the files the microfacet_tensorf2 path uses:

    configs/default.yaml                  top-level keys + `defaults: [_self_, dataset: X, model: Y, field: Z]`
    configs/<group>/<name>.yaml           group files, mounted under the key <group>
    cfg.model.arch.rf = cfg.field         (train.py:911)

compose() implements the defaults-list composition, `group=name` choices, dotted `a.b.c=value` overrides (values parsed as
YAML, `1e-3` is a float as in OmegaConf) and comma sweeps (`-m`: one config per combination).  instantiate_arch() maps the
`_target_` strings of the reference's packages onto this package's operator classes and applies
the `_partial_` semantics of hydra.utils.instantiate.  dump() writes the resolved config like train.py:485.
"""
import copy
import functools
import itertools
import os
import re

import yaml

# OmegaConf / YAML 1.2 floats: PyYAML (YAML 1.1) reads `1e-3` as a string
_FLOAT = re.compile(r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                        |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                        |\.[0-9_]+(?:[eE][-+][0-9]+)?
                        |[-+]?\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$""", re.X)


class _Loader(yaml.SafeLoader):
    pass


_Loader.add_implicit_resolver("tag:yaml.org,2002:float", _FLOAT, list("-+0123456789."))


def _load(path):
    with open(path) as f:
        return yaml.load(f, Loader=_Loader) or {}


def _parse_value(text):
    return yaml.load(text, Loader=_Loader)


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set(cfg, dotted, value):
    keys = dotted.split(".")
    node = cfg
    for k in keys[:-1]:
        if not isinstance(node.get(k), dict):
            node[k] = {}
        node = node[k]
    node[keys[-1]] = value


def _reader(config_dir):
    """relative path without extension -> dict: a configs/ directory, or (config_dir None) the built-in tree of
    nmf_amd/config.py -- the same files as values, so the package composes without the reference checkout"""
    if config_dir is not None:
        def read(rel):
            path = os.path.join(config_dir, rel + ".yaml")
            if not os.path.exists(path):
                raise FileNotFoundError(f"no config file {path}")
            node = _load(path)
            if not isinstance(node, dict):
                raise ValueError(f"{path}: a config file must hold a mapping")
            return node
        return read
    from .config import builtin_tree
    tree = builtin_tree()

    def read(rel):
        if rel not in tree:
            have = sorted(k.split("/", 1)[1] for k in tree if k.startswith(rel.split("/")[0] + "/"))
            raise FileNotFoundError(f"{rel} is not part of the built-in config tree (have: {have}); pass --config-dir "
                                    "to compose from a configs/ directory")
        return copy.deepcopy(tree[rel])
    return read


def compose(config_dir, overrides=(), config_name="default"):
    """-> resolved config dict (one run).  overrides: ["model=microfacet_tensorf2", "dataset=lego", "expname=x",
    "model.arch.model.anoise=0.1", ...].  config_dir None: the built-in tree."""
    read = _reader(config_dir)
    base = read(config_name)
    defaults = base.pop("defaults", ["_self_"])
    groups = [next(iter(d)) for d in defaults if isinstance(d, dict)]
    choice, dotted = {}, []
    for ov in overrides:
        key, _, val = ov.partition("=")
        key = key.lstrip("+")
        if key in groups and "." not in key:
            choice[key] = val
        else:
            dotted.append((key, _parse_value(val)))
    cfg = {}
    for d in defaults:
        if d == "_self_":
            _merge(cfg, base)
        else:
            group, name = next(iter(d.items()))
            name = choice.get(group, name)
            if name in (None, "null"):
                continue
            cfg[group] = _merge(cfg.get(group, {}) if isinstance(cfg.get(group), dict) else {}, read(f"{group}/{name}"))
    if "_self_" not in defaults:
        _merge(cfg, base)
    for key, val in dotted:
        _set(cfg, key, val)
    if "field" in cfg and isinstance(cfg.get("model"), dict) and "arch" in cfg["model"]:
        cfg["model"]["arch"]["rf"] = copy.deepcopy(cfg["field"])                       # train.py:911
    return cfg


def sweep(config_dir, overrides=(), config_name="default"):
    """hydra -m: every comma-separated override value spans one axis -> list of (overrides, config)."""
    axes = []
    for ov in overrides:
        key, _, val = ov.partition("=")
        vals = [v for v in val.split(",")] if ("," in val and not val.strip().startswith("[")) else [val]
        axes.append([f"{key}={v}" for v in vals])
    return [(list(combo), compose(config_dir, combo, config_name)) for combo in itertools.product(*axes)]


# `_target_` of the reference -> class of this package
def _targets():
    from .brdf_samplers.ggx import GGXSampler
    from .fields.tensoRF import TensorVMSplit
    from .models.microfacet import Microfacet
    from .modules.brdf import MLPBRDF, ListISH
    from .modules.integral_equirect import IntegralEquirect
    from .modules.render_modules import RandHydraMLPDiffuse
    from .modules.tensor_nerf import TensorNeRF
    from .modules.tonemap import SRGBTonemap
    from .samplers.alphagrid import AlphaGridSampler
    return {"modules.tensor_nerf.TensorNeRF": TensorNeRF, "modules.tonemap.SRGBTonemap": SRGBTonemap,
            "samplers.alphagrid.AlphaGridSampler": AlphaGridSampler, "models.microfacet.Microfacet": Microfacet,
            "brdf_samplers.ggx.GGXSampler": GGXSampler, "modules.brdf.MLPBRDF": MLPBRDF, "modules.ish.ListISH": ListISH,
            "modules.render_modules.RandHydraMLPDiffuse": RandHydraMLPDiffuse,
            "modules.integral_equirect.IntegralEquirect": IntegralEquirect, "fields.tensoRF.TensorVMSplit": TensorVMSplit}


def instantiate(node, targets=None):
    """hydra.utils.instantiate for plain dicts: `_target_` -> call (or functools.partial when `_partial_`), recursively."""
    targets = targets if targets is not None else _targets()
    if isinstance(node, dict):
        if "_target_" in node:
            name = node["_target_"]
            if name not in targets:
                raise NotImplementedError(f"_target_ {name} is outside the microfacet_tensorf2 path (SURVEY section 8)")
            kwargs = {k: instantiate(v, targets) for k, v in node.items() if k not in ("_target_", "_partial_")}
            cls = targets[name]
            return functools.partial(cls, **kwargs) if node.get("_partial_", False) else cls(**kwargs)
        return {k: instantiate(v, targets) for k, v in node.items()}
    if isinstance(node, list):
        return [instantiate(v, targets) for v in node]
    return node


def instantiate_arch(cfg, aabb, near_far):
    """train.py:239-247: tensorf = hydra.utils.instantiate(cfg.model.arch)(aabb=..., near_far=...)"""
    return instantiate(cfg["model"]["arch"])(aabb=aabb, near_far=list(near_far))


def dump(cfg, path):
    """train.py:485 (OmegaConf.save)"""
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f, sort_keys=False)
