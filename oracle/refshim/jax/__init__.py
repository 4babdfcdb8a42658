"""NumPy stand-in for the slice of `jax` the reference's dense path uses (oracle/refshim/README.md).

TEST INFRASTRUCTURE ONLY.  float64 throughout (= JAX with jax_enable_x64)."""
from __future__ import annotations

import types

import numpy as _np

from . import numpy  # noqa: F401  (jax.numpy)
from . import scipy  # noqa: F401  (jax.scipy.linalg)
from ._array import Array, wrap
from . import tree_util  # noqa: F401

__version__ = "0.0-refshim"


def jit(fun=None, **_static):
    """Identity: there is nothing to trace."""
    if fun is None:
        return lambda f: f
    return fun


def _index(tree, i, axis):
    return tree_util.tree_map(lambda x: wrap(_np.take(_np.asarray(x), i, axis=axis)), tree)


def vmap(fun, in_axes=0, out_axes=0):
    """Loop over the mapped axis and stack: the same per-point arithmetic JAX batches."""

    def mapped(*args):
        axes = tuple(in_axes) if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
        if len(axes) != len(args):
            raise ValueError("vmap: in_axes must match the positional arguments")
        n = None
        for a, ax in zip(args, axes):
            if ax is None:
                continue
            for leaf in tree_util.tree_leaves(a):
                m = _np.shape(leaf)[ax]
                if n is not None and m != n:
                    raise ValueError("vmap: inconsistent sizes along the mapped axes")
                n = m
        if n is None:
            raise ValueError("vmap: nothing to map over")
        if n == 0:
            raise ValueError("vmap (refshim): empty mapped axis")
        outs = [fun(*[a if ax is None else _index(a, i, ax) for a, ax in zip(args, axes)])
                for i in range(n)]
        return tree_util.tree_map(lambda *xs: wrap(_np.stack([_np.asarray(x) for x in xs], axis=out_axes)),
                                  *outs)

    return mapped


class _Config:
    def update(self, *_a, **_k):
        pass


config = _Config()


def _stub_module(name):
    mod = types.ModuleType(name)

    def __getattr__(attr):
        def _raise(*_a, **_k):
            raise NotImplementedError(f"{name}.{attr} is outside the dense DirectSolver path (refshim)")

        return _raise

    mod.__getattr__ = __getattr__
    return mod


lax = _stub_module("jax.lax")
random = _stub_module("jax.random")
debug = _stub_module("jax.debug")
