"""Config surface: the hand-resolved dictionaries of nmf_amd/config.py and the YAML composer nmf_amd/yaml_config.py against
the values the reference's own YAML files resolve to (tests/golden/config_resolved.json, written by
tests/golden/make_config_fixture.py from /root/reference/configs)."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "config_resolved.json")


def _yc():
    spec = importlib.util.spec_from_file_location("yaml_config", os.path.join(ROOT, "nmf_amd", "yaml_config.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _leaves(node, prefix=""):
    if isinstance(node, dict):
        for k, v in node.items():
            yield from _leaves(v, f"{prefix}.{k}" if prefix else k)
    else:
        yield prefix, node


def test_builtin_config_matches_reference_yaml_values():
    """every hyper-parameter hard-wired in nmf_amd/config.py equals what the reference's YAML files resolve to"""
    spec = importlib.util.spec_from_file_location("cfgmod", os.path.join(ROOT, "nmf_amd", "config.py"))
    src = open(os.path.join(ROOT, "nmf_amd", "config.py")).read()
    ns = {}
    exec(src.split("def instantiate")[0], ns)                   # MODEL / FIELD / resolved_config without the kernels
    mine = ns["resolved_config"]()
    ref = json.load(open(GOLD))
    ref_arch, ref_params = ref["model"]["arch"], ref["model"]["params"]
    ref_leaf = dict(_leaves(ref_arch))
    checked = 0
    for path, v in _leaves(mine["arch"]):
        if path.endswith("_target_"):
            assert ref_leaf[path].split(".")[-1] == v.split(".")[-1], path
            continue
        if path == "rf" and v == "placeholder":
            continue
        assert path in ref_leaf, f"{path} is not a key of the reference config"
        r = ref_leaf[path]
        assert (r == v) or (isinstance(r, (int, float)) and isinstance(v, (int, float)) and float(r) == float(v)), (path, v, r)
        checked += 1
    for k, v in mine["params"].items():
        assert k in ref_params and (ref_params[k] == v or float(ref_params[k]) == float(v)), (k, v, ref_params.get(k))
        checked += 1
    assert checked > 80


def test_yaml_composer_semantics(tmp_path):
    yc = _yc()
    d = tmp_path / "configs"
    (d / "model").mkdir(parents=True)
    (d / "field").mkdir()
    (d / "default.yaml").write_text("defaults:\n  - _self_\n  - model: a\n  - field: f\nseed: 1\nlr: 1e-3\nckpt: NULL\n")
    (d / "model" / "a.yaml").write_text("arch:\n  _target_: modules.tonemap.SRGBTonemap\n  rf: placeholder\nparams:\n  eps: 1e-8\n  betas: [0.9, 0.99]\n")
    (d / "model" / "b.yaml").write_text("arch:\n  rf: placeholder\n  k: 2\nparams:\n  eps: 1\n")
    (d / "field" / "f.yaml").write_text("_target_: fields.tensoRF.TensorVMSplit\ngrid_size: [128, 128, 128]\n")
    cfg = yc.compose(str(d))
    assert cfg["lr"] == 1e-3 and isinstance(cfg["lr"], float) and cfg["ckpt"] is None and cfg["seed"] == 1
    assert cfg["model"]["params"]["eps"] == 1e-8 and cfg["model"]["arch"]["rf"]["grid_size"] == [128] * 3   # train.py:911
    cfg = yc.compose(str(d), ["model=b", "seed=7", "model.params.eps=2.5e-4", "model.arch.new.key=[1,2]"])
    assert cfg["model"]["arch"]["k"] == 2 and cfg["seed"] == 7 and cfg["model"]["params"]["eps"] == 2.5e-4
    assert cfg["model"]["arch"]["new"]["key"] == [1, 2]
    runs = yc.sweep(str(d), ["model=a,b", "seed=1,2,3"])
    assert len(runs) == 6 and {r[1]["seed"] for r in runs} == {1, 2, 3}
    out = tmp_path / "config.yaml"
    yc.dump(cfg, str(out))
    assert yc._load(str(out))["model"]["params"]["eps"] == 2.5e-4


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference checkout not present")
def test_composer_on_the_reference_tree():
    yc = _yc()
    cfg = yc.compose("/root/reference/configs", ["model=microfacet_tensorf2", "field=tensorf_og", "dataset=lego"])
    assert cfg == json.load(open(GOLD))
    assert cfg["dataset"]["near_far"] == [2.5, 7] and cfg["model"]["arch"]["rf"]["_target_"] == "fields.tensoRF.TensorVMSplit"
