"""Configuration surface: the reference's ``config/*.yaml`` files are read unchanged.

The reference flattens every YAML section onto one namespace at import time from ``sys.argv``
(util/config.py:11-40).  Here the same flattening is a function, so a harness can load a yaml
explicitly; ``cfg`` is the process-wide namespace the model classes fall back to (same role as
``util.config.cfg``), and ``python x.py --config path.yaml`` still works through ``from_argv``.
"""
from __future__ import annotations

import argparse
import os
from types import SimpleNamespace

import yaml

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CONFIG_DIR = os.path.join(REPO_ROOT, "config")

cfg = SimpleNamespace()


def load_config(path: str, **overrides) -> SimpleNamespace:
    """Flatten every section of the yaml into one namespace (util/config.py:27-34)."""
    if not os.path.isabs(path) and not os.path.exists(path):
        path = os.path.join(CONFIG_DIR, path)
    with open(path, "r") as f:
        sections = yaml.safe_load(f)
    ns = SimpleNamespace(config=path, pretrain=None, resume=None, output_path=None, local_rank=0)
    for sec in sections.values():
        for k, v in sec.items():
            setattr(ns, k, v)
    for k, v in overrides.items():
        setattr(ns, k, v)
    ns.exp_path = ns.output_path
    return ns


def set_config(path_or_ns, **overrides) -> SimpleNamespace:
    """Install a configuration as the global ``cfg`` (in place, so earlier imports see it)."""
    ns = load_config(path_or_ns, **overrides) if isinstance(path_or_ns, str) else path_or_ns
    cfg.__dict__.clear()
    cfg.__dict__.update(ns.__dict__)
    return cfg


def from_argv(argv=None) -> SimpleNamespace:
    p = argparse.ArgumentParser(description="GeoFormer (MI355X build)")
    p.add_argument("--config", type=str, default="geoformer_scannet.yaml")
    p.add_argument("--pretrain", type=str, default=None)
    p.add_argument("--resume", type=str, default=None)
    p.add_argument("--output_path", type=str, default=None)
    p.add_argument("--local_rank", type=int, default=0)
    a, _ = p.parse_known_args(argv)
    return set_config(a.config, pretrain=a.pretrain, resume=a.resume, output_path=a.output_path,
                      local_rank=a.local_rank)
