"""ndarray subclass with JAX's functional `.at[idx].add/.set` (reference noise.py:77-78)."""
import numpy as np


class _At:
    def __init__(self, arr):
        self._arr = arr

    def __getitem__(self, idx):
        return _AtIdx(self._arr, idx)


class _AtIdx:
    def __init__(self, arr, idx):
        self._arr, self._idx = arr, idx

    def add(self, v):
        out = np.array(self._arr, copy=True)
        np.add.at(out, self._idx, v)
        return out.view(Array)

    def set(self, v):
        out = np.array(self._arr, copy=True)
        out[self._idx] = v
        return out.view(Array)


class Array(np.ndarray):
    @property
    def at(self):
        return _At(self)


def wrap(x):
    if isinstance(x, np.ndarray):
        return x if isinstance(x, Array) else x.view(Array)
    if isinstance(x, (np.floating, np.integer, np.bool_, np.complexfloating)):
        return np.asarray(x).view(Array)
    if isinstance(x, (tuple, list)):
        return type(x)(wrap(v) for v in x)
    return x
