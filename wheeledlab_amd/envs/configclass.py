"""`configclass`: the small subset of IsaacLab's `isaaclab.utils.configclass` the reference's configs rely on
(class attributes become per-instance fields with deep-copied defaults, keyword construction, `replace`, `to_dict`,
`__post_init__`).  Reference usage: e.g. wheeledlab_tasks/drifting/mushr_drift_env_cfg.py:38,78,242,368."""
from __future__ import annotations

import copy
import types

MISSING = type("MISSING", (), {"__repr__": lambda s: "MISSING", "__deepcopy__": lambda s, m: s, "__copy__": lambda s: s})()


def _is_field(name, value, owner=None):
    if name.startswith("__"):
        return False
    if isinstance(value, (types.FunctionType, classmethod, staticmethod, property)):
        return False
    if isinstance(value, type):  # nested class definitions are not fields; `class_type = SomeTerm` is
        return not (owner is not None and value.__qualname__.startswith(owner.__qualname__ + "."))
    return True


def _collect_fields(cls):
    fields = {}
    for klass in reversed(cls.__mro__):
        if klass is object:
            continue
        ann = klass.__dict__.get("__annotations__", {})
        for name in ann:
            if name not in klass.__dict__ and not name.startswith("__"):
                fields.setdefault(name, MISSING)
        for name, value in klass.__dict__.items():
            if _is_field(name, value, klass):
                fields[name] = value
    return fields


def configclass(cls):
    fields = _collect_fields(cls)
    cls.__cfg_fields__ = fields
    user_post = cls.__dict__.get("__post_init__")

    def __init__(self, **kwargs):
        for name, default in type(self).__cfg_fields__.items():
            setattr(self, name, default if isinstance(default, type) else copy.deepcopy(default))
        for k, v in kwargs.items():
            if k not in type(self).__cfg_fields__:
                raise TypeError(f"{type(self).__name__} has no config field '{k}'")
            setattr(self, k, v)
        post = getattr(self, "__post_init__", None)
        if post is not None:
            post()

    def replace(self, **kwargs):
        new = copy.deepcopy(self)
        for k, v in kwargs.items():
            setattr(new, k, v)
        return new

    def to_dict(self):
        def plain(v):
            if isinstance(v, type) or (callable(v) and not hasattr(v, "to_dict")):   # a class / function held as a value
                return f"{getattr(v, '__module__', '')}:{getattr(v, '__qualname__', repr(v))}"
            if hasattr(v, "to_dict"):
                return v.to_dict()
            if isinstance(v, dict):
                return {kk: plain(vv) for kk, vv in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(plain(x) for x in v)
            return v
        return {k: plain(v) for k, v in self.__dict__.items()}

    def __repr__(self):
        body = ", ".join(f"{k}={v!r}" for k, v in self.__dict__.items())
        return f"{type(self).__name__}({body})"

    cls.__init__ = __init__
    cls.replace = replace
    cls.to_dict = to_dict
    cls.__repr__ = __repr__
    if user_post is None and not any("__post_init__" in k.__dict__ for k in cls.__mro__[1:]):
        cls.__post_init__ = lambda self: None
    return cls


def fields_of(cfg):
    """(name, value) pairs of a config instance in declaration order, then attributes set on the INSTANCE afterwards
    (`cfg.rewards.my_term = RewTerm(...)`: IsaacLab's managers iterate the instance dict, so such terms count)"""
    declared = type(cfg).__cfg_fields__
    names = [k for k in declared if hasattr(cfg, k)] + [k for k in vars(cfg) if k not in declared and not k.startswith("_")]
    return [(k, getattr(cfg, k)) for k in names]
