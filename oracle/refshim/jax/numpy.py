"""`jax.numpy` -> NumPy, float64 defaults, results wrapped in `Array` (refshim)."""
import numpy as _np

from ._array import Array, wrap

pi, inf, nan, e = _np.pi, _np.inf, _np.nan, _np.e
float32, float64, int32, int64, bool_ = _np.float32, _np.float64, _np.int32, _np.int64, _np.bool_
ndarray = Array


def finfo(x):
    return _np.finfo(getattr(x, "dtype", x))


def asarray(x, dtype=None):
    return wrap(_np.asarray(x, dtype=dtype))


array = asarray


def ndim(x):
    return _np.ndim(x)


def shape(x):
    return _np.shape(x)


def __getattr__(name):
    fn = getattr(_np, name)
    if not callable(fn) or isinstance(fn, type):
        return fn

    def wrapped(*a, **k):
        return wrap(fn(*a, **k))

    wrapped.__name__ = name
    return wrapped
