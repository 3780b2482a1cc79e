"""YAML config surface of `build_centernet()` (Gen-A nested layout of the reference's configs/*.yaml).

Reads only the `model:` section (reference README.md:29-37, configs/base_resnet34_fpn.yaml:1-25); `data:`,
`trainer:`, optimizer / lr_scheduler / loss keys are tolerated and ignored.  Supports `__base__`
inheritance (configs/helmet.yaml:1) and both spellings of neck kwargs: directly under `neck:`
(configs/base_resnet34.yaml:7-11) or nested under `neck.params` (configs/test_config.yaml:8-18).
The reference's own loader (`load_config`, referenced at models/tracker.py:12) is missing from its tree.
"""
import copy
import os
from typing import Any, Dict, Union

import yaml


def _deep_update(base: dict, new: dict) -> dict:
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            _deep_update(base[k], v)
        else:
            base[k] = copy.deepcopy(v)
    return base


def load_config(config: Union[str, Dict[str, Any]]) -> Dict[str, Any]:
    """Return the full config dict with `__base__` files merged in (child overrides parent)."""
    if isinstance(config, dict):
        return copy.deepcopy(config)
    path = os.fspath(config)
    with open(path, "r") as f:
        cfg = yaml.safe_load(f) or {}
    base = cfg.pop("__base__", None)
    if base is not None:
        base_path = base if os.path.isabs(base) else os.path.join(os.path.dirname(path), base)
        merged = load_config(base_path)
        return _deep_update(merged, cfg)
    return cfg


def model_section(config: Union[str, Dict[str, Any]]) -> Dict[str, Any]:
    cfg = load_config(config)
    model = cfg["model"] if "model" in cfg else cfg          # accept the bare model dict too
    for key in ("backbone", "neck", "output_heads"):
        if key not in model:
            raise KeyError(f"config has no model.{key}")
    model = copy.deepcopy(model)
    neck = model["neck"]
    if isinstance(neck.get("params"), dict):                  # test_config.yaml nesting
        params = neck.pop("params")
        for k, v in params.items():
            neck.setdefault(k, v)
    model.setdefault("task", "tracking" if "reid" in model["output_heads"] else "detection")
    return model
