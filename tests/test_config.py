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


def _leaf_eq(a, b):
    return a == b or (isinstance(a, (int, float)) and isinstance(b, (int, float)) and not isinstance(a, bool)
                      and not isinstance(b, bool) and float(a) == float(b))


def _compare(mine, ref, skip=()):
    lm, lr = dict(_leaves(mine)), dict(_leaves(ref))
    assert sorted(lm) == sorted(lr), (sorted(set(lm) - set(lr)), sorted(set(lr) - set(lm)))
    for k, v in lr.items():
        if k in skip:
            continue
        assert _leaf_eq(lm[k], v), (k, lm[k], v)
    return len(lr)


HYDRA = os.path.join(ROOT, "tests", "golden", "hydra_config_car.json")


def test_composer_reproduces_a_config_hydra_itself_wrote():
    """The pin the composer did not produce: /root/reference/config.yaml was written by hydra / OmegaConf (train.py:485) for a
    dataset=car model=microfacet_tensorf2 field=tensorf_og run.  Composing the same choices + the overrides that undo what the YAML
    files have changed since gives the same key set and every leaf (the three biases the run calibrated are written at run time).
    Runs from the built-in tree (no reference checkout needed) and, where the checkout exists, from the reference's own files."""
    sys_path_root()
    from nmf_amd import yaml_config as yc
    fx = json.load(open(HYDRA))
    n = _compare(yc.compose(None, fx["overrides"]), fx["config"], skip=fx["run_time_leaves"])
    assert n > 150
    if os.path.isdir("/root/reference/configs"):
        _compare(yc.compose("/root/reference/configs", fx["overrides"]), fx["config"], skip=fx["run_time_leaves"])


def sys_path_root():
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)


def test_builtin_tree_composes_what_the_reference_files_compose():
    """`python -m nmf_amd.train dataset=lego ...` without a configs/ directory: the built-in tree (nmf_amd/config.py) resolves to the
    fixture of the reference's files for lego, and for every built-in scene to what the reference's files give here."""
    sys_path_root()
    from nmf_amd import yaml_config as yc
    from nmf_amd.config import DATASETS
    assert yc.compose(None, ["dataset=lego"]) == json.load(open(GOLD))
    assert yc.compose(None, ["dataset=lego", "model=microfacet_tensorf2", "field=tensorf_og"]) == json.load(open(GOLD))
    with pytest.raises(FileNotFoundError):
        yc.compose(None, ["model=brdf_tcnn"])                    # outside the path: named, not silently replaced
    if os.path.isdir("/root/reference/configs"):
        for name in DATASETS:
            if name == "s2_orbit":
                continue
            ov = [f"dataset={name}", "model=microfacet_tensorf2", "field=tensorf_og"]
            assert yc.compose(None, ov) == yc.compose("/root/reference/configs", ov), name


def test_train_cli_composes_hydra_tokens_and_flags():
    """the command line of train.py:904-921: group choices, dotted overrides, and the flag shorthands land in the same tree"""
    sys_path_root()
    import types
    from nmf_amd import train as T
    a = types.SimpleNamespace(datadir=None, near_far=None, downsample=1.0, grid=None, bg=None, seed=None, views=None, test_views=None,
                              res=None, config_dir=None)
    cfg = T.compose_run(a, ["model=microfacet_tensorf2", "field=tensorf_og", "dataset=helmet", "model.arch.model.anoise=0.1",
                            "model.params.n_iters=500", "expname=x"])
    assert cfg["dataset"]["near_far"] == [3, 5] and cfg["dataset"]["aabb_scale"] == 2 and cfg["expname"] == "x"
    assert cfg["model"]["arch"]["model"]["anoise"] == 0.1 and cfg["model"]["params"]["n_iters"] == 500
    assert cfg["model"]["arch"]["rf"] == cfg["field"] and cfg["field"]["grid_size"] == [128, 128, 128]
    assert T.compose_run(a, [])["dataset"]["dataset_name"] == "synthetic_orbit"         # no data directory: the offline stand-in
    a.datadir, a.grid, a.bg, a.near_far = "/tmp/scenes/lego", 16, 32, [2.0, 6.0]
    cfg = T.compose_run(a, [])
    assert os.path.join(cfg["datadir"], cfg["dataset"]["scenedir"]) == "/tmp/scenes/lego" and cfg["dataset"]["near_far"] == [2.0, 6.0]
    assert cfg["model"]["arch"]["rf"]["grid_size"] == [16, 16, 16] and cfg["model"]["arch"]["bg_module"]["bg_resolution"] == 32


def test_datadir_flag_is_not_parsed_as_yaml_and_multirun_axes():
    """`--datadir /data/007` must open /data/007 (the YAML value parser reads '007' as 7, 'yes' as True, 'a: b' as a
    mapping); `-m` sweeps are the product of the comma-separated values in hydra's order (README.md:10)."""
    import argparse
    from nmf_amd import train as T
    from nmf_amd import yaml_config as yc
    ns = argparse.Namespace(datadir="/data/007", near_far=None, downsample=1.0, grid=None, bg=None, seed=None, views=None,
                            test_views=None, res=None, config_dir=None)
    cfg = T.compose_run(ns, [])
    assert cfg["datadir"] == "/data" and cfg["dataset"]["scenedir"] == "007" and cfg["dataset"]["dataset_name"] == "blender"
    for name in ("yes", "1e3", "0x10", "null", "a: b", "[x"):
        ns.datadir = "/data/" + name
        assert T.compose_run(ns, [])["dataset"]["scenedir"] == name
    jobs = [c for c, _ in yc.sweep(None, ["expname=v", "dataset=ficus,drums,ship", "model.arch.model.rays_per_ray=16,32"])]
    assert len(jobs) == 6 and jobs[0] == ["expname=v", "dataset=ficus", "model.arch.model.rays_per_ray=16"]
    assert jobs[1][2].endswith("=32") and jobs[2][1] == "dataset=drums"
