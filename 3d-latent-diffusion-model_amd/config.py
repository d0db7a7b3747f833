"""``define_instance``: the reference's plug point (3d_ldm/utils.py:243-246) without MONAI's ConfigParser.

The reference merges two JSON files into an argparse Namespace (3d_ldm/train_diffusion.py:58-64,
3d_ldm/inference.py:61-67) and asks MONAI's ConfigParser to instantiate one key.  The shipped configs use
exactly three syntactic features: ``"_target_"``, ``"@name"`` references and ``"$expr"``
Python expressions that may contain ``@name``.  This resolver implements those and maps the MONAI class paths
the configs name onto the MI355X-native classes.
"""
from __future__ import annotations

import importlib
import json
import re
from types import SimpleNamespace
from typing import Any, Dict

# dotted class paths the reference's configs use -> our implementations
TARGET_ALIASES = {
    "networks.DiffusionModelUNet": "ldm3d.networks.DiffusionModelUNet",
    "networks.AutoencoderKL": "ldm3d.networks.AutoencoderKL",
    "monai.networks.nets.DiffusionModelUNet": "ldm3d.networks.DiffusionModelUNet",
    "monai.networks.nets.AutoencoderKL": "ldm3d.networks.AutoencoderKL",
    "generative.networks.nets.DiffusionModelUNet": "ldm3d.networks.DiffusionModelUNet",
    "generative.networks.nets.AutoencoderKL": "ldm3d.networks.AutoencoderKL",
    "monai.networks.schedulers.DDPMScheduler": "ldm3d.schedulers.DDPMScheduler",
    "monai.networks.schedulers.DDIMScheduler": "ldm3d.schedulers.DDIMScheduler",
    "generative.networks.schedulers.DDPMScheduler": "ldm3d.schedulers.DDPMScheduler",
    "generative.networks.schedulers.DDIMScheduler": "ldm3d.schedulers.DDIMScheduler",
    "monai.inferers.LatentDiffusionInferer": "ldm3d.inferer.LatentDiffusionInferer",
    "generative.inferers.LatentDiffusionInferer": "ldm3d.inferer.LatentDiffusionInferer",
}

_REF = re.compile(r"@([A-Za-z_][A-Za-z0-9_]*(?:(?:::|#)[A-Za-z0-9_]+)*)")


class ConfigResolver:
    def __init__(self, config: Dict[str, Any]):
        self.config = config
        self._resolving: set = set()

    def _lookup(self, ref: str):
        node: Any = self.config
        for part in re.split(r"::|#", ref):
            if isinstance(node, (list, tuple)):
                node = node[int(part)]
            elif part in node:
                node = node[part]
            else:
                raise KeyError(f"config reference '@{ref}' not found")
        if ref in self._resolving:
            raise ValueError(f"circular config reference '@{ref}'")
        self._resolving.add(ref)
        try:
            return self.resolve(node, instantiate=True)
        finally:
            self._resolving.discard(ref)

    def resolve(self, node: Any, instantiate: bool = True):
        if isinstance(node, dict):
            out = {k: self.resolve(v, instantiate) for k, v in node.items() if not k.startswith("_") or k == "_target_"}
            if instantiate and "_target_" in out:
                if node.get("_disabled_", False):
                    return None
                return instantiate_target(out.pop("_target_"), out)
            return out
        if isinstance(node, (list, tuple)):
            return [self.resolve(v, instantiate) for v in node]
        if isinstance(node, str):
            if node.startswith("$"):
                # "$@image_channels", "$[@a, 2*@b]": substitute refs by local names, then eval
                local: Dict[str, Any] = {}

                def sub(m):
                    name = "__ref_%d" % len(local)
                    local[name] = self._lookup(m.group(1))
                    return name
                expr = _REF.sub(sub, node[1:])
                return eval(expr, {"__builtins__": {"len": len, "int": int, "float": float, "max": max, "min": min,
                                                    "range": range, "list": list, "tuple": tuple, "sum": sum}}, local)
            m = _REF.fullmatch(node)
            if m:
                return self._lookup(m.group(1))
        return node

    def get_parsed_content(self, key: str, instantiate: bool = True):
        return self.resolve(self.config[key], instantiate)


def instantiate_target(target: str, kwargs: Dict[str, Any]):
    path = TARGET_ALIASES.get(target, target)
    mod_name, _, cls_name = path.rpartition(".")
    try:
        cls = getattr(importlib.import_module(mod_name), cls_name)
    except (ImportError, AttributeError) as e:
        raise ImportError(f"cannot resolve _target_ '{target}' (-> '{path}'): {e}") from e
    return cls(**kwargs)


def define_instance(args, instance_def_key: str):
    """Same signature and behaviour as 3d_ldm/utils.py:243-246: ``args`` is the Namespace the JSON files were
    merged into (or a plain dict)."""
    cfg = dict(vars(args)) if not isinstance(args, dict) else args
    return ConfigResolver(cfg).get_parsed_content(instance_def_key, instantiate=True)


def load_config_namespace(environment_file: str, config_file: str, **extra) -> SimpleNamespace:
    """JSON -> Namespace merge used by every entry script (3d_ldm/inference.py:61-67)."""
    ns = SimpleNamespace(**extra)
    for path in (environment_file, config_file):
        if path:
            with open(path, "r") as f:
                for k, v in json.load(f).items():
                    setattr(ns, k, v)
    return ns
