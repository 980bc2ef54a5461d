"""YAML config loading with `${key}` interpolation across files, as scripts/train.py does through the
third-party `metayaml.read` (/root/reference/scripts/train.py:43-51; e.g.
configs/experiments/celeba-10pts.yaml:15-16 reference ${logdir}, ${name}, ${celeba_data_dir},
${vgg16_path} from configs/paths/default.yaml).  Files are merged left to right (later files win,
dicts merge recursively); substitution resolves dotted paths against the merged tree.
"""
import re

import yaml

from .box import Box

_VAR = re.compile(r'\$\{([^}]+)\}')


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _lookup(tree, path):
    cur = tree
    for part in path.split('.'):
        cur = cur[part]
    return cur


def _subst(node, tree, depth=0):
    if depth > 16:
        raise ValueError('config: ${} substitution does not terminate')
    if isinstance(node, dict):
        return {k: _subst(v, tree, depth) for k, v in node.items()}
    if isinstance(node, list):
        return [_subst(v, tree, depth) for v in node]
    if isinstance(node, str):
        m = _VAR.fullmatch(node)
        if m:   # whole-string reference keeps the referenced type
            return _subst(_lookup(tree, m.group(1).strip()), tree, depth + 1)
        if _VAR.search(node):
            return _subst(_VAR.sub(lambda mm: str(_lookup(tree, mm.group(1).strip())), node), tree, depth + 1)
    return node


def load_configs(file_names):
    """Counterpart of scripts/train.py:43-51 load_configs."""
    tree = {}
    for fn in file_names:
        with open(fn, 'r') as f:
            data = yaml.safe_load(f) or {}
        _merge(tree, data)
    return Box(_subst(tree, tree))
