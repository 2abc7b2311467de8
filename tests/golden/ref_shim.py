"""Import shim that makes the read-only reference at /root/reference importable.

TEST INFRASTRUCTURE ONLY.  Used by ``gen_golden.py`` in the build container to
produce the committed golden vectors; nothing in the product, the ``-m gpu``
tests, ``smoke()`` or ``bench.py`` imports this (``/root/reference`` does not
exist on the GPU box).

The reference imports, at module-import time, packages that are absent here
(hydra_zen, lightning, lightning_utilities, torchmetrics, timm and, through
``mmlearn/datasets/__init__.py``, torchvision/cv2).  We register minimal
stand-in modules for exactly the names the imports need; the
arithmetic on the hot path is stock torch, except
``torchmetrics.utilities.compute._safe_matmul`` (torchmetrics 1.6.2, not
vendored, "parity unpinned"): restated from its documented contract as
``x @ y.T`` with an fp32 up-cast for fp16 inputs.
"""

from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

REFERENCE_ROOT = os.environ.get("MMLEARN_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "mmlearn"))


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def install() -> None:
    """Install the stand-in modules and put the reference on sys.path."""
    if "mmlearn" in sys.modules and getattr(sys.modules["mmlearn"], "_shimmed", False):
        return
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")

    import torch
    from torch import nn

    # transformers probes optional deps through importlib specs; import it
    # before any spec-less stand-in exists.
    import transformers  # noqa: F401
    import transformers.modeling_outputs  # noqa: F401
    import transformers.tokenization_utils_base  # noqa: F401

    # ---- hydra_zen -------------------------------------------------------
    def store(*args, **kwargs):
        if args and (callable(args[0]) or isinstance(args[0], type)) and len(args) == 1:
            return args[0]
        if args:  # store(obj, name=..., group=...)
            return args[0]

        def deco(obj):
            return obj

        return deco

    class _ZenStore:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return store(*a, **k)

        def add_to_hydra_store(self, *a, **k):
            pass

    def builds(*a, **k):
        return SimpleNamespace(args=a, kwargs=k)

    _mod("hydra_zen", store=store, MISSING="???", builds=builds, ZenStore=_ZenStore)

    # ---- lightning ---------------------------------------------------------
    class LightningModule(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self.logged = {}
            self.trainer = SimpleNamespace(sanity_checking=False)
            self._device = torch.device("cpu")

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, name, value, **kw):
            self.logged[name] = value

        @property
        def device(self):
            return self._device

    def rank_zero_warn(*a, **k):
        pass

    def move_data_to_device(batch, device):
        return batch

    L = _mod("lightning", LightningModule=LightningModule)
    _mod("lightning.pytorch", LightningModule=LightningModule)
    _mod("lightning.pytorch.utilities", move_data_to_device=move_data_to_device)
    _mod("lightning.pytorch.utilities.types", OptimizerLRScheduler=object)
    _mod("lightning.pytorch.utilities.rank_zero", rank_zero_warn=rank_zero_warn)
    _mod("lightning.fabric")
    _mod("lightning.fabric.utilities", rank_zero_warn=rank_zero_warn)
    L.pytorch = sys.modules["lightning.pytorch"]

    _mod("lightning_utilities")
    _mod("lightning_utilities.core")
    _mod("lightning_utilities.core.rank_zero", rank_zero_warn=rank_zero_warn)

    class RequirementCache:
        def __init__(self, *a, **k):
            pass

        def __bool__(self):
            return False

    _mod("lightning_utilities.core.imports", RequirementCache=RequirementCache)

    # ---- torchmetrics ------------------------------------------------------
    class Metric(nn.Module):
        """Stand-in for torchmetrics.Metric: just enough state handling for the reference's metrics to run on one process
        (add_state keeps a copy of the default as an attribute; no sync, no reset bookkeeping)."""

        def __init__(self, *a, **k):
            super().__init__()
            self.process_group = None
            self.distributed_available_fn = lambda: False

        def add_state(self, name, default, dist_reduce_fx=None, persistent=False):
            setattr(self, name, [] if isinstance(default, list) else default.clone())

    class _Dummy(Metric):
        pass

    def _safe_matmul(x, y):
        if x.dtype == torch.float16 or y.dtype == torch.float16:
            return (x.float() @ y.T.float()).half()
        return x @ y.T

    _mod(
        "torchmetrics",
        Metric=Metric,
        MetricCollection=_Dummy,
        AUROC=_Dummy,
        Accuracy=_Dummy,
        F1Score=_Dummy,
        Precision=_Dummy,
        Recall=_Dummy,
    )
    _mod("torchmetrics.utilities")
    _mod("torchmetrics.utilities.compute", _safe_matmul=_safe_matmul)
    _mod("torchmetrics.utilities.checks", _check_same_shape=lambda *a, **k: None)
    _mod("torchmetrics.utilities.data", dim_zero_cat=lambda x: torch.cat(list(x), 0))
    _mod("torchmetrics.utilities.distributed", gather_all_tensors=lambda x, *a, **k: [x])
    _mod("torchmetrics.retrieval")
    def _retrieval_aggregate(values, aggregation="mean", dim=None):
        # torchmetrics 1.6.2 retrieval/base.py (not vendored): mean / median / min / max over the values, or a callable
        if aggregation == "mean":
            return values.mean() if dim is None else values.mean(dim=dim)
        if aggregation == "median":
            return values.median() if dim is None else values.median(dim=dim).values
        if aggregation == "min":
            return values.min() if dim is None else values.min(dim=dim).values
        if aggregation == "max":
            return values.max() if dim is None else values.max(dim=dim).values
        return aggregation(values, dim=dim)

    _mod("torchmetrics.retrieval.base", _retrieval_aggregate=_retrieval_aggregate)

    # ---- timm --------------------------------------------------------------
    def global_pool_nlc(x, pool_type="", num_prefix_tokens=1, reduce_include_prefix=False):
        if not pool_type:
            return x
        if pool_type == "token":
            return x[:, 0]
        x = x if reduce_include_prefix else x[:, num_prefix_tokens:]
        if pool_type == "avg":
            return x.mean(dim=1)
        if pool_type == "max":
            return x.amax(dim=1)
        if pool_type == "avgmax":
            return 0.5 * (x.amax(dim=1) + x.mean(dim=1))
        raise ValueError(pool_type)

    class _TimmViT(nn.Module):
        pass

    _mod("timm", create_model=lambda *a, **k: None)
    _mod("timm.models")
    _mod(
        "timm.models.vision_transformer",
        VisionTransformer=_TimmViT,
        global_pool_nlc=global_pool_nlc,
    )

    # ---- the reference itself ---------------------------------------------
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # bypass mmlearn/datasets/__init__.py (imports torchvision / cv2 datasets)
    ds = types.ModuleType("mmlearn.datasets")
    ds.__path__ = [os.path.join(REFERENCE_ROOT, "mmlearn", "datasets")]
    import mmlearn  # noqa: F401  (package __init__ is light)

    sys.modules["mmlearn.datasets"] = ds
    sys.modules["mmlearn"].datasets = ds
    sys.modules["mmlearn"]._shimmed = True


def load():
    """Return a namespace of the reference symbols on the hot path."""
    install()
    from mmlearn.datasets.core import find_matching_indices
    from mmlearn.datasets.core.modalities import Modalities
    from mmlearn.datasets.processors.masking import IJEPAMaskGenerator, apply_masks
    from mmlearn.datasets.processors.transforms import repeat_interleave_batch
    from mmlearn.modules.ema import ExponentialMovingAverage
    from mmlearn.modules.encoders.vision import (
        VisionTransformer,
        VisionTransformerPredictor,
    )
    from mmlearn.modules.layers.logit_scaling import LearnableLogitScaling
    from mmlearn.modules.layers.normalization import L2Norm
    from mmlearn.modules.losses.contrastive import ContrastiveLoss
    from mmlearn.tasks.contrastive_pretraining import (
        ContrastivePretraining,
        LossPairSpec,
        ModuleKeySpec,
    )
    from mmlearn.tasks.ijepa import IJEPA

    return SimpleNamespace(**{k: v for k, v in locals().items()})
