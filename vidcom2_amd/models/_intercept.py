"""Plumbing shared by the model hooks.

The reference's hooks (token_compressor/vidcom2/models/*.py) are whole-function copies of upstream
LLaVA-NeXT / HF transformers methods with one or two inserted calls into the compression pass.
The hooks here are *wrappers* instead: they run the model class's own, unmodified method and
intercept the two or three attributes on the instance that bracket the insertion point.  They
therefore follow whatever upstream version is installed and carry no upstream code.

`shadow(obj, name=value, ...)` puts `value` into the instance `__dict__` for the duration of a
`with` block.  An instance-dict entry wins over class functions (non-data descriptors) and over
`nn.Module` sub-modules / parameters (those are only reached through `__getattr__`), so no module
state is touched and everything is restored on exit, also when the wrapped call raises.
"""
from __future__ import annotations

import contextlib
import os
from typing import Any, Callable

_MISSING = object()


@contextlib.contextmanager
def shadow(obj: Any, **attrs: Any):
    saved = {}
    d = obj.__dict__
    try:
        for name, value in attrs.items():
            saved[name] = d.get(name, _MISSING)
            d[name] = value
        yield
    finally:
        for name, old in saved.items():
            if old is _MISSING:
                d.pop(name, None)
            else:
                d[name] = old


def original_method(obj: Any, name: str, hook: Callable) -> Callable:
    """The class's own `name`, bound to `obj` -- i.e. what ran before `hook` was installed with
    `types.MethodType(hook, obj)` (README.md:76-83 of the reference) or assigned on a subclass."""
    for klass in type(obj).__mro__:
        fn = klass.__dict__.get(name)
        if fn is None:
            continue
        if getattr(fn, "__func__", fn) is hook:
            continue
        return fn.__get__(obj, type(obj))
    raise AttributeError(f"{type(obj).__name__} has no original '{name}' to wrap "
                         "(the hook must be installed on a model that defines it)")


def retention_ratio() -> float:
    """`R_RATIO` env knob, default 0.25 (models/llava.py:124, models/qwen2_5_vl.py:135)."""
    return float(os.getenv("R_RATIO", "0.25"))


def compressor_enabled() -> bool:
    """`COMPRESSOR=vidcom2` env knob (models/qwen2_5_vl.py:121)."""
    return os.getenv("COMPRESSOR") == "vidcom2"
