"""The slice of the reference's `utils` the hot path uses (utils/__init__.py:81-92, 99-114)."""
import json
import os
from collections import namedtuple

from hipops import ops


def norm(x):
    """[0,1] -> [-1,1], in place on the caller's tensor like upstream (utils/__init__.py:88-92)."""
    return ops.affine_(x, 2.0, -1.0)


def denorm(x, vmin=0, vmax=1):
    """[-1,1] -> [vmin,vmax], in place (utils/__init__.py:81-86)."""
    return ops.affine_(x, 0.5 * (vmax - vmin), 0.5 * (vmax - vmin) + vmin)


def load_json(path):
    """JSON -> nested namedtuple; JSON `false` becomes None, as upstream does (utils/__init__.py:99-106)."""
    def hook(d):
        d = {k: (None if v is False else v) for k, v in d.items()}
        return namedtuple('X', d.keys())(*d.values())
    with open(path) as f:
        return json.load(f, object_hook=hook)


def get_world_size():
    return int(os.environ.get('WORLD_SIZE', 1))


def is_distributed():
    return get_world_size() > 1
