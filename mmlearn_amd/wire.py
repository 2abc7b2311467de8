"""Batch / id wire format: the host side that feeds ``example_ids`` to the contrastive loss.

Mirrors of the reference's ``Example`` (mmlearn/datasets/core/example.py:11-99), ``collate_example_list`` /
``DefaultDataCollator`` (datasets/core/data_collator.py:13-137) and ``CombinedDataset``
(datasets/core/combined_dataset.py:14-129): same keys, tensors and errors.  On top of the reference's batch the collator
emits, while the ids are still host memory (no device work, no sync):

* ``batch["fully_paired"]`` -- ``True`` when every modality of the batch carries the same id column in the same order.
  ``ContrastiveLoss.forward(..., fully_paired=True)`` then pairs row ``p`` with row ``p`` directly: no ``match_ids`` launch,
  no status read-back (the one host sync of the loss path) and, across ranks, no id all-gather.
* ``batch["example_keys"][modality]`` -- int64 ``[B]`` keys ``dataset_index << 32 | example_index``: the same identity in
  half the bytes; ``pack_example_ids`` / ``unpack_example_keys`` convert, and the loss accepts either form.

Both additions are derived data: a consumer that ignores them sees the reference's batch.
"""

from __future__ import annotations

import bisect
import warnings
from collections import OrderedDict
from collections.abc import Mapping, MutableMapping
from dataclasses import dataclass
from itertools import accumulate
from typing import Any, Callable, Hashable, Iterable, Optional

import torch
from torch.utils._pytree import tree_flatten
from torch.utils.data import Dataset, IterableDataset, default_collate

from .modalities import Modalities

_ID_FIELDS = ("example_ids", "example_index", "dataset_index")


class Example(OrderedDict):
    """One sample: an ordered mapping with attribute access (``ex.text`` is ``ex["text"]``); mappings assigned after
    construction become ``Example``s themselves (example.py:84-98)."""

    def __init__(self, init_dict: Optional[MutableMapping[Hashable, Any]] = None) -> None:
        super().__init__({} if init_dict is None else init_dict)

    def create_ids(self) -> None:
        """``example_ids[key] = tensor([dataset_index, example_index])`` for every payload key (example.py:41-77)."""
        if "example_index" not in self or "dataset_index" not in self:
            warnings.warn("Cannot create `example_ids` without `example_index` and `dataset_index` attributes. "
                          "Set these attributes before calling `create_ids`. No `example_ids` was created.",
                          category=UserWarning, stacklevel=2)
            return
        pair = [self["dataset_index"], self["example_index"]]
        self.example_ids = {key: torch.tensor(pair) for key in list(self.keys()) if key not in _ID_FIELDS}

    def __getattr__(self, key: str) -> Any:
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    def __setattr__(self, key: str, value: Any) -> None:
        self[key] = value

    def __setitem__(self, key: Hashable, value: Any) -> None:
        super().__setitem__(key, Example(value) if isinstance(value, MutableMapping) else value)


# ---------------------------------------------------------------------------------------------------------------------
def pack_example_ids(example_ids: torch.Tensor) -> torch.Tensor:
    """``[N, 2]`` (dataset_index, example_index) -> int64 ``[N]`` keys ``dataset_index << 32 | example_index``."""
    if example_ids.dim() != 2 or example_ids.shape[1] != 2:
        raise ValueError(f"example ids must be [N, 2], got {tuple(example_ids.shape)}")
    ids = example_ids.to(torch.int64)
    if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= 1 << 32):
        raise ValueError("dataset_index / example_index must fit 32 unsigned bits to be packed")
    return (ids[:, 0] << 32) | ids[:, 1]


def unpack_example_keys(keys: torch.Tensor) -> torch.Tensor:
    """Inverse of :func:`pack_example_ids`: int64 ``[N]`` -> ``[N, 2]``."""
    if keys.dim() != 1:
        raise ValueError(f"example keys must be [N], got {tuple(keys.shape)}")
    keys = keys.to(torch.int64)
    return torch.stack([(keys >> 32) & 0xFFFFFFFF, keys & 0xFFFFFFFF], dim=1)


def _merge(examples: Iterable[Mapping]) -> dict[str, Any]:
    """Per key, the list of the values of the samples that have it; nested Examples merge recursively
    (data_collator.py:83-110)."""
    columns: dict[str, Any] = {}
    for ex in examples:
        for key, value in ex.items():
            columns.setdefault(key, []).append(value)
    return {k: _merge(v) if isinstance(v[0], Example) else v for k, v in columns.items()}


def _collate(columns: Mapping[str, Any]) -> dict[str, Any]:
    return {k: _collate(v) if isinstance(v, dict) else default_collate(v) for k, v in columns.items()}


def collate_example_list(examples: list) -> dict[str, Any]:
    """``default_collate`` per key over the samples that carry the key (data_collator.py:65-137)."""
    return _collate(_merge(examples))


def pairing_summary(example_ids: Mapping[str, torch.Tensor]) -> tuple[bool, dict[str, torch.Tensor]]:
    """(fully_paired, packed keys) of a collated ``example_ids`` dict; host tensors, no device work.  ``fully_paired``
    needs the common id column to be duplicate-free as well: for a repeated id ``find_matching_indices`` yields every
    (i, j) combination, more pairs than the identity pairing the flag stands for."""
    keys, first, paired = {}, None, len(example_ids) > 0
    for name, ids in example_ids.items():
        if not (isinstance(ids, torch.Tensor) and ids.dim() == 2 and ids.shape[1] == 2):
            return False, {}
        k = pack_example_ids(ids.cpu())
        keys[name] = k
        if first is None:
            first = k
        elif paired:
            paired = k.shape == first.shape and bool(torch.equal(k, first))
    if paired and first is not None and first.numel() and torch.unique(first).numel() != first.numel():
        paired = False
    return paired, keys


@dataclass
class DefaultDataCollator:
    """Collate ``Example``s, then run ``batch_processors[key]`` on ``batch[key]`` (modality names resolve to the
    modality's batch key); a processor returning a mapping must contain the key and is merged into the batch
    (data_collator.py:13-62).  Adds ``fully_paired`` / ``example_keys`` (module docstring) unless ``wire_format=False``."""

    batch_processors: Optional[dict[str, Callable[[Any], Any]]] = None
    wire_format: bool = True

    def __call__(self, examples: list) -> dict[str, Any]:
        batch = collate_example_list(examples)
        for key, fn in (self.batch_processors or {}).items():
            batch_key = Modalities.get_modality(key).name if Modalities.has_modality(key) else key
            if batch_key not in batch:
                continue
            out = fn(batch[batch_key])
            if isinstance(out, Mapping):
                if batch_key not in out:
                    raise ValueError(f"Batch processor for '{key}' key must return a dictionary with '{batch_key}' in it.")
                batch.update(out)
            else:
                batch[batch_key] = out
        if self.wire_format and isinstance(batch.get("example_ids"), Mapping):
            paired, keys = pairing_summary(batch["example_ids"])
            if keys:
                batch["fully_paired"], batch["example_keys"] = paired, keys
        return batch


# ---------------------------------------------------------------------------------------------------------------------
class CombinedDataset(Dataset):
    """Concatenation of map-style and sized iterable-style datasets (combined_dataset.py:14-129): a global index is
    located by bisection over the cumulative sizes; iterable members are read sequentially and restarted when
    exhausted; every sample gets ``dataset_index`` and ``example_ids`` unless it already has them."""

    def __init__(self, datasets: Iterable) -> None:
        self.datasets, _ = tree_flatten(datasets)
        if any(not isinstance(d, (Dataset, IterableDataset)) for d in self.datasets):
            raise TypeError("Expected argument `datasets` to be an iterable of `Dataset` or `IterableDataset` instances, "
                            f"but found: {self.datasets}")
        if not self.datasets:
            raise ValueError("Expected a non-empty iterable of datasets but found an empty iterable")
        self._cumulative_sizes = list(accumulate(len(d) for d in self.datasets))
        self._streams = {i: iter(d) for i, d in enumerate(self.datasets) if isinstance(d, IterableDataset)}

    def __len__(self) -> int:
        return self._cumulative_sizes[-1]

    def __getitem__(self, idx: int) -> Example:
        n = len(self)
        if idx < 0:
            if -idx > n:
                raise IndexError(f"Index {idx} is out of bounds for the combined dataset with length {n}")
            idx += n
        which = bisect.bisect_right(self._cumulative_sizes, idx)
        member = self.datasets[which]
        if which in self._streams:
            try:
                sample = next(self._streams[which])
            except StopIteration:
                self._streams[which] = iter(member)
                sample = next(self._streams[which])
        else:
            sample = member[idx - (self._cumulative_sizes[which - 1] if which else 0)]
        if not isinstance(sample, Example):
            raise TypeError(f"Expected dataset examples to be instances of `Example` but found {type(sample)}")
        if "dataset_index" not in sample:
            sample.dataset_index = which
        if "example_ids" not in sample:
            sample.create_ids()
        return sample
