"""Minimal Hydra/OmegaConf stand-in for the lid launchers (neither package is in this image; real Hydra is used when it is
importable).  Supports what the reference's confs use: ``defaults: - group: name`` includes, YAML anchors,
``${a.b.c}`` interpolation, ``--config-name``, dotted ``key=value`` / ``+key=value`` overrides and a per-run output dir."""
import argparse
import datetime
import functools
import os
import re
import sys

import yaml


class Cfg(dict):
    """dict with attribute access (enough of DictConfig for ``cfg["trainer"]`` / ``**cfg.model`` style use)."""
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return self.get(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _lookup(root, dotted):
    cur = root
    for part in dotted.split("."):
        cur = cur[int(part)] if isinstance(cur, list) else cur[part]
    return cur


def _interpolate(node, root, depth=0):
    if isinstance(node, dict):
        return {k: _interpolate(v, root, depth) for k, v in node.items()}
    if isinstance(node, list):
        return [_interpolate(v, root, depth) for v in node]
    if isinstance(node, str) and "${" in node and depth < 8:
        def sub(m):
            key = m.group(1)
            if key.startswith("now:"):
                return datetime.datetime.now().strftime(key[4:])
            return str(_interpolate(_lookup(root, key), root, depth + 1))
        whole = re.fullmatch(r"\$\{([^}]+)\}", node)
        if whole and not whole.group(1).startswith("now:"):
            return _interpolate(_lookup(root, whole.group(1)), root, depth + 1)
        return re.sub(r"\$\{([^}]+)\}", sub, node)
    return node


def _set(root, dotted, value):
    parts = dotted.split(".")
    cur = root
    for p in parts[:-1]:
        cur = cur.setdefault(p, {}) if isinstance(cur, dict) else cur[int(p)]
    cur[parts[-1]] = value


def load_config(config_dir, config_name, overrides=()):
    with open(os.path.join(config_dir, config_name + ".yaml")) as f:
        cfg = yaml.safe_load(f) or {}
    for entry in cfg.pop("defaults", []) or []:
        if isinstance(entry, dict):
            for group, name in entry.items():
                path = os.path.join(config_dir, group, f"{name}.yaml")
                if os.path.exists(path):
                    with open(path) as f:
                        base = yaml.safe_load(f) or {}
                    cfg = {**base, **cfg}
    for ov in overrides:
        key, _, val = ov.lstrip("+").partition("=")
        _set(cfg, key, yaml.safe_load(val))
    return _wrap(_interpolate(cfg, cfg))


def main(config_path="conf", config_name="config"):
    """Decorator with ``@hydra.main``'s call shape."""
    def deco(fn):
        @functools.wraps(fn)
        def run():
            ap = argparse.ArgumentParser()
            ap.add_argument("--config-name", default=config_name)
            ap.add_argument("overrides", nargs="*")
            a = ap.parse_args()
            here = os.path.dirname(os.path.abspath(sys.modules[fn.__module__].__file__))
            cfg = load_config(os.path.join(here, config_path), a.config_name, a.overrides)
            run_dir = (cfg.get("hydra") or {}).get("run", {}).get("dir") or os.path.join(
                "outputs", datetime.datetime.now().strftime("%Y-%m-%d/%H-%M-%S"))
            os.makedirs(run_dir, exist_ok=True)
            os.chdir(run_dir)
            return fn(cfg)
        return run
    return deco
