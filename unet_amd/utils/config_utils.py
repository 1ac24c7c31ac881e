"""YAML config loading with the reference's override and number-coercion behaviour
(/root/reference/utils/config_utils.py:5-44): `--over a.b=v` values are Python-evaluated when they
parse, and strings such as "1e-4" (which PyYAML leaves as str) become floats."""
import ast
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


def _coerce_numbers(obj):
    if isinstance(obj, dict):
        return {k: _coerce_numbers(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_coerce_numbers(v) for v in obj]
    if isinstance(obj, str):
        s = obj.strip()
        if s.startswith("(") and s.endswith(")"):
            try:
                a, b = [float(t) for t in s[1:-1].split(",")]
                return (a, b)
            except Exception:
                return obj
        if s.isdigit():
            return int(s)
        try:
            return float(s)
        except ValueError:
            return obj
    return obj


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
