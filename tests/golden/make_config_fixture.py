"""Writes tests/golden/config_resolved.json: the VALUES the reference's YAML files resolve to for
`model=microfacet_tensorf2 field=tensorf_og dataset=lego` (configs/default.yaml composition + train.py:911), produced by
nmf_amd/yaml_config.py reading /root/reference/configs in this container.  Data only (hyper-parameter values).
    python tests/golden/make_config_fixture.py
"""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("yaml_config", os.path.join(HERE, "..", "..", "nmf_amd", "yaml_config.py"))
yc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(yc)

cfg = yc.compose("/root/reference/configs", ["model=microfacet_tensorf2", "field=tensorf_og", "dataset=lego"])
with open(os.path.join(HERE, "config_resolved.json"), "w") as f:
    json.dump(cfg, f, indent=1, sort_keys=True)
print("wrote config_resolved.json:", len(json.dumps(cfg)), "bytes")
