"""Modality names and the ``<name>_embedding`` key convention of the boundary.

The loss receives ``embeddings`` keyed by ``Modality.embedding`` and ``example_ids`` keyed by
``Modality.name`` (mmlearn/modules/losses/contrastive.py:257-258; datasets/core/modalities.py:15-300).
When mmlearn itself is importable its registry singleton is used, so modalities registered
by a project are visible here; otherwise an equivalent minimal registry is provided.
"""

from __future__ import annotations

import re
import warnings
from dataclasses import dataclass, field
from typing import Optional

try:  # drop-in deployment: share the reference's registry
    from mmlearn.datasets.core.modalities import Modalities, Modality  # type: ignore # noqa: F401

    USING_MMLEARN_REGISTRY = True
except Exception:  # standalone
    USING_MMLEARN_REGISTRY = False

    _DEFAULT_SUPPORTED_MODALITIES = ["rgb", "depth", "thermal", "text", "audio", "video"]
    _DEFAULT_PROPERTIES = ["target", "attention_mask", "mask", "embedding", "masked_embedding", "ema_embedding"]

    def _is_format_string(s: str) -> bool:
        return bool(re.search(r"\{.*?\}", s))

    @dataclass
    class Modality:
        name: str
        modality_specific_properties: Optional[dict] = field(default=None, repr=False)

        def __post_init__(self) -> None:
            self.name = self.name.lower()
            self._properties: dict[str, str] = {}
            for prop in _DEFAULT_PROPERTIES:
                self._properties[prop] = f"{self.name}_{prop}"
                setattr(self, prop, self._properties[prop])
            for k, fmt in (self.modality_specific_properties or {}).items():
                self.add_property(k, fmt)

        @property
        def properties(self) -> dict[str, str]:
            return self._properties

        def add_property(self, name: str, format_string: str) -> None:
            if name in self._properties:
                warnings.warn(f"Property '{name}' already exists for modality '{self.name}'. Will overwrite the existing property.",
                              category=UserWarning, stacklevel=2)
            if not _is_format_string(format_string):
                raise ValueError(f"Invalid format string '{format_string}' for property '{name}' of modality '{self.name}'.")
            self._properties[name] = format_string.format(self.name)
            setattr(self, name, self._properties[name])

        def __str__(self) -> str:
            return self.name.lower()

        def __hash__(self) -> int:
            return hash(self.name)

    class ModalityRegistry:
        _instance = None

        def __new__(cls):
            if cls._instance is None:
                cls._instance = super().__new__(cls)
                cls._instance._registry = {}
            return cls._instance

        def register_modality(self, name: str, modality_specific_properties: Optional[dict] = None) -> None:
            if name.lower() in self._registry:
                warnings.warn(f"Modality '{name}' already exists in the registry. Overwriting...", category=UserWarning, stacklevel=2)
            self._registry[name.lower()] = Modality(name, modality_specific_properties)

        def add_default_property(self, name: str, format_string: str) -> None:
            for m in self._registry.values():
                m.add_property(name, format_string)

        def has_modality(self, name: str) -> bool:
            return name.lower() in self._registry

        def get_modality(self, name: str) -> Modality:
            return self._registry[name.lower()]

        def get_modality_properties(self, name: str) -> dict:
            return self.get_modality(name).properties

        def list_modalities(self) -> list:
            return list(self._registry.values())

        def __getattr__(self, name: str) -> Modality:
            reg = self.__dict__.get("_registry", {})
            if name.lower() in reg:
                return reg[name.lower()]
            raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    Modalities = ModalityRegistry()
    for _m in _DEFAULT_SUPPORTED_MODALITIES:
        if not Modalities.has_modality(_m):
            Modalities.register_modality(_m)
