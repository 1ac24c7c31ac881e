"""YAML config loading with the reference's override and number-coercion behaviour
(/root/reference/utils/config_utils.py:5-44): `--over a.b=v` values are Python-evaluated when they
parse, and strings such as "1e-4" (which PyYAML leaves as str) become floats."""
import ast
import re
from typing import Dict, List

import yaml


def _parse_value(v: str):
    try:
        return ast.literal_eval(v)          # the reference uses eval(); literals cover its use
    except Exception:
        try:
            return eval(v, {"__builtins__": {}}, {})
        except Exception:
            return v


# What PyYAML leaves as a string although the configs mean a number (it only resolves floats that contain a dot, so
# "1e-4" stays text) -- the behaviour of the reference's loader (utils/config_utils.py:24-44), as a rule table:
# first matching rule converts, a rule that raises leaves the string as it is.
_PAIR = re.compile(r"^\((.*)\)$")
_RULES = (
    (lambda s: _PAIR.match(s) is not None, lambda s: tuple(float(t) for t in _split_pair(s))),    # "(0.9, 0.999)" -> (0.9, 0.999)
    (str.isdigit, int),                                                                          # "32" -> 32
    (lambda s: True, float),                                                                     # "1e-4" -> 0.0001
)


def _split_pair(s: str):
    parts = _PAIR.match(s).group(1).split(",")
    if len(parts) != 2:
        raise ValueError(s)
    return parts


def _coerce_scalar(text: str):
    s = text.strip()
    for applies, convert in _RULES:
        if applies(s):
            try:
                return convert(s)
            except ValueError:
                return text
    return text


def _coerce_numbers(node):
    """Apply _coerce_scalar to every string leaf of a nested dict / list config."""
    if isinstance(node, str):
        return _coerce_scalar(node)
    if isinstance(node, dict):
        return {k: _coerce_numbers(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_coerce_numbers(v) for v in node]
    return node


def load_cfg(path: str, overrides: List[str] = None) -> Dict:
    with open(path, "r") as f:
        cfg = yaml.safe_load(f)
    for kv in overrides or []:
        key, val = kv.split("=", 1)
        cur = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = _parse_value(val)
    return _coerce_numbers(cfg)
