"""One data-plumbing scenario, run once with the reference's ``Example`` / ``CombinedDataset`` / ``DefaultDataCollator``
(by gen_golden.py, which commits the outputs as g11_wire.npz) and once with ``mmlearn_amd.wire`` (by the tests).
TEST INFRASTRUCTURE ONLY.  The scenario is this repo's own; only the classes under test are passed in."""

from collections.abc import Mapping

import numpy as np
import torch
from torch.utils.data import Dataset, IterableDataset


def _flatten(prefix, obj, out):
    if isinstance(obj, Mapping):
        out[prefix + "@keys"] = np.array(list(obj.keys()))   # key ORDER is part of the contract
        for k, v in obj.items():
            _flatten(f"{prefix}{k}.", v, out)
    elif isinstance(obj, torch.Tensor):
        out[prefix + "t"] = obj.numpy()
        out[prefix + "dtype"] = np.array(str(obj.dtype))
    elif isinstance(obj, (list, tuple)) and obj and isinstance(obj[0], str):
        out[prefix + "s"] = np.array(list(obj))
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            _flatten(f"{prefix}{i}.", v, out)
    else:
        out[prefix + "v"] = np.array(obj)


def run(Example, CombinedDataset, DefaultDataCollator):
    g = torch.Generator().manual_seed(11)
    rgb = torch.randn(8, 3, 2, 2, generator=g)
    tok = torch.randint(0, 50, (9, 4), generator=g)
    wav = torch.randn(3, 6, generator=g)

    class Pairs(Dataset):          # rgb + text, map-style
        def __len__(self):
            return 5

        def __getitem__(self, i):
            return Example({"rgb": rgb[i], "text": tok[i], "caption": f"cap{i}", "example_index": i})

    class TextOnly(Dataset):       # text + a nested mapping
        def __len__(self):
            return 4

        def __getitem__(self, i):
            return Example({"text": tok[5 + i], "meta": {"len": torch.tensor(i + 1), "src": {"page": i * 10}}, "example_index": i})

    class Stream(IterableDataset):  # rgb + audio, iterable with a length; restarts when exhausted
        def __len__(self):
            return 3

        def __iter__(self):
            for i in range(3):
                yield Example({"rgb": rgb[5 + i], "audio": wav[i], "example_index": i})

    ds = CombinedDataset([Pairs(), [TextOnly(), Stream()]])   # nested containers are flattened
    out = {"len": np.array(len(ds))}
    order = [0, 4, 5, 8, 9, 10, 11, 9, -1, -12, 2]
    samples = [ds[i] for i in order]
    for j, s in enumerate(samples):
        _flatten(f"sample{j}.", s, out)

    def text_proc(t):
        return {"text": t + 1, "attention_mask": (t > 10).long()}

    def rgb_proc(x):
        return x * 2.0

    batch = DefaultDataCollator(batch_processors={"text": text_proc, "rgb": rgb_proc, "depth": rgb_proc})(samples)
    batch = {k: v for k, v in batch.items() if k not in ("fully_paired", "example_keys")}   # derived additions, tested apart
    _flatten("batch.", batch, out)
    paired = DefaultDataCollator()([ds[i] for i in (3, 1, 2)])
    paired = {k: v for k, v in paired.items() if k not in ("fully_paired", "example_keys")}
    _flatten("paired.", paired, out)
    return out
