"""Writes the two config fixtures (data only: hyper-parameter values).
    python tests/golden/make_config_fixture.py

config_resolved.json   what the reference's YAML files resolve to for `model=microfacet_tensorf2 field=tensorf_og dataset=lego`
                       (configs/default.yaml composition + train.py:911), composed by nmf_amd/yaml_config.py reading
                       /root/reference/configs in this container.
hydra_config_car.json  /root/reference/config.yaml -- a resolved config written by HYDRA ITSELF (OmegaConf.save, train.py:485) for a
                       `dataset=car model=microfacet_tensorf2 field=tensorf_og` run of the reference's authors -- parsed to JSON, next
                       to the command-line overrides that reproduce it from today's YAML files (the field file has changed since:
                       grid 64 -> 128, density_shift -10 -> -4, `numer_grad` dropped, ...).  This is the pin of the composer that
                       the composer did not produce: tests/test_config.py composes those overrides and compares every leaf.
"""
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
spec = importlib.util.spec_from_file_location("yaml_config", os.path.join(HERE, "..", "..", "nmf_amd", "yaml_config.py"))
yc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(yc)

cfg = yc.compose("/root/reference/configs", ["model=microfacet_tensorf2", "field=tensorf_og", "dataset=lego"])
with open(os.path.join(HERE, "config_resolved.json"), "w") as f:
    json.dump(cfg, f, indent=1, sort_keys=True)
print("wrote config_resolved.json:", len(json.dumps(cfg)), "bytes")

hydra = yc._load("/root/reference/config.yaml")
overrides = ["dataset=car", "model=microfacet_tensorf2", "field=tensorf_og", "expname=v1_neural_sm1", "datadir=/optane/nerf_datasets",
             "lr_decay_iters=30000", "model.params.final_pred_lambda=0.0003", "field.grid_size=[64,64,64]", "field.density_shift=-10",
             "field.N_voxel_init=262144", "field.upsamp_list=[500,1000,2000,3000,4000,5500,7000]", "+field.numer_grad=true"]
# written by the run itself, not by the composition (train.py:429-437 stores the calibrated biases in the config before saving)
run_time = ["model.arch.model.brdf.bias", "model.arch.model.diffuse_module.diffuse_bias",
            "model.arch.model.diffuse_module.roughness_bias"]
with open(os.path.join(HERE, "hydra_config_car.json"), "w") as f:
    json.dump({"source": "/root/reference/config.yaml (OmegaConf.save of a hydra-composed run)", "overrides": overrides,
               "run_time_leaves": run_time, "config": hydra}, f, indent=1, sort_keys=True)
print("wrote hydra_config_car.json")
