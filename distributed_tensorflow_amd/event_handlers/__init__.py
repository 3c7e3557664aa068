"""Event-handler plug-ins for the job watcher (reference ``eventHandlers/``).

The reference's directory has no ``__init__.py``, so the watcher's dynamic
``importlib.import_module('eventHandlers.'+name)`` (``service/tf_cleaner.py:28``)
cannot work as shipped.  Here handlers are importable both by
module path and through :func:`load_handler`.
"""
import importlib


def load_handler(name):
    """``"CleanJobHandler"`` -> class, looked up as
    ``event_handlers.<Name>.<Name>`` like the reference; dotted names
    (``pkg.mod.Class``) select third-party plug-ins."""
    name = name.strip()
    if "." in name:
        mod, cls = name.rsplit(".", 1)
        return getattr(importlib.import_module(mod), cls)
    m = importlib.import_module(__name__ + "." + name)
    return getattr(m, name)
