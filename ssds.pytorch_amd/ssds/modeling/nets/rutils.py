"""``@register`` puts a backbone factory into its module's ``__all__`` so that
``getattr(nets, cfg.MODEL.NETS)`` finds it (reference ``ssds/modeling/nets/rutils.py:4-9``)."""
import sys


def register(f):
    names = sys.modules[f.__module__].__dict__.setdefault("__all__", [])
    if f.__name__ in names:
        raise RuntimeError("{} already exist!".format(f.__name__))
    names.append(f.__name__)
    return f
