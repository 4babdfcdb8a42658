"""`equinox` -> dataclasses (refshim): Module = dataclass applied by a metaclass, with
`__check_init__` hooks and abstract-method enforcement; `field(static=..., converter=...)`
keeps only the default.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import abc
import dataclasses

__version__ = "0.0-refshim"


def field(*, default=dataclasses.MISSING, default_factory=dataclasses.MISSING, static=False,
          converter=None, init=True, **_ignored):
    kw = {"init": init, "metadata": {"static": static}}
    if default is not dataclasses.MISSING:
        kw["default"] = default
    if default_factory is not dataclasses.MISSING:
        kw["default_factory"] = default_factory
    return dataclasses.field(**kw)


class _ModuleMeta(abc.ABCMeta):
    def __new__(mcs, name, bases, ns, **kw):
        cls = super().__new__(mcs, name, bases, ns, **kw)
        own_init = "__init__" in ns
        inherited_custom = any(getattr(b, "_eqx_custom_init", False) for b in bases)
        cls._eqx_custom_init = own_init or inherited_custom
        # a class that inherits a hand-written __init__ keeps it (equinox does the same)
        return dataclasses.dataclass(init=not (inherited_custom and not own_init), eq=False,
                                     repr=False)(cls)

    def __call__(cls, *args, **kwargs):
        obj = super().__call__(*args, **kwargs)
        for klass in reversed(cls.__mro__):
            check = klass.__dict__.get("__check_init__")
            if check is not None:
                check(obj)
        return obj


class Module(metaclass=_ModuleMeta):
    pass
