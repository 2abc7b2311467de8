"""Registration under mmlearn's hydra-zen store groups, import-guarded.

mmlearn registers configurable classes with ``@store(group=..., provider="mmlearn")`` on the
global hydra-zen store (losses/contrastive.py:19, tasks/contrastive_pretraining.py:87,
tasks/ijepa.py:24) and lets out-of-tree code use ``mmlearn.conf.external_store``
(conf/__init__.py:190).  Here:

* mmlearn importable  -> register through ``mmlearn.conf.external_store`` (provider "mmlearn_amd");
* only hydra-zen      -> register on ``hydra_zen.store`` under the same group names;
* neither (this image)-> ``store`` is an identity decorator, recorded in ``REGISTERED`` so that
  the wiring stays testable.

YAML use (drop-in):  ``/modules/losses@task.loss: ContrastiveLossHIP`` / ``override /task: ContrastivePretrainingHIP``.
"""

from __future__ import annotations

from typing import Any, Callable

REGISTERED: dict[tuple[str, str], Any] = {}
BACKEND = "none"

try:
    from mmlearn.conf import external_store as _store  # type: ignore

    BACKEND = "mmlearn.external_store"
except Exception:
    try:
        from hydra_zen import store as _hz_store  # type: ignore

        _store = _hz_store
        BACKEND = "hydra_zen.store"
    except Exception:
        _store = None


def store(*args: Any, **kwargs: Any) -> Callable:
    """``@store(group="modules/losses", name=...)`` with mmlearn's calling conventions."""

    def _register(obj: Any, **kw: Any) -> Any:
        name = kw.get("name", getattr(obj, "__name__", str(obj)))
        REGISTERED[(kw.get("group", ""), name)] = obj
        if _store is not None:
            kw.setdefault("provider", "mmlearn_amd")
            _store(obj, **kw)
        return obj

    if args and callable(args[0]) and not kwargs:
        return _register(args[0])
    if args:
        return _register(args[0], **kwargs)
    return lambda obj: _register(obj, **kwargs)
